#!/bin/bash
# round 6: GPU tests, C3 by kernel, the n_cu/2 < n_sc < n_cu shape question, the box's CPU share, the default bench line
O=gpurun_out/${1:-r6c}; mkdir -p $O
set -o pipefail
python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt; [ $rc -eq 0 ] || exit $rc
python tools/experiments/r05_c3_kernels.py > $O/c3_kernels.txt 2>&1 && cat $O/c3_kernels.txt
for n in 40 80 96 112 127; do
  for env in "" "NAE_DEBUG=pv_fps=4"; do
    env $env python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-pcie --no-host-path --sustain-seconds 0 --total-streams $n | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('streams %4d  [%-12s] step %.3f ms | ' % ($n, '$env', d['ms_per_step']) + ' '.join('%s %.3f' % (k.replace('_kernel',''), v['avg_ms']) for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_ms'])))
"
  done
done > $O/shape_96.txt 2>&1; cat $O/shape_96.txt
python -c "
import os
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for p in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpuset.cpus.effective'):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, 'n/a')
" > $O/cpus.txt 2>&1; cat $O/cpus.txt
( time python bench.py ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python -c "
import json
d=json.loads(open('$O/bench.json').read())
print('step', d['ms_per_step'], 'cpu_baseline', d['cpu_baseline'])
"
