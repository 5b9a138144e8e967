#!/bin/bash
# one-box A/B of NAE_PV_LOCKSTEP=N: step time, pv_synth time and its WRITE_SIZE (KiB per launch; algorithmic = 3 840 000)
cd /tmp; export TMPDIR=/tmp
for v in ${@:-0 1 4 8 16}; do
  if [ $v != 0 ]; then export NAE_PV_LOCKSTEP=$v; else unset NAE_PV_LOCKSTEP; fi
  echo "== lockstep $v"
  python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  step %.2f ms | ' % d['ms_per_step'] + ' '.join('%s %.2f' % (k.replace('_kernel',''), v['avg_ms']) for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_ms'])[:5]))
"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/lw$v -o w -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-alt --steps 2 --warmup 1 > /tmp/lw$v.log 2>&1
  python - <<PY
import csv, collections
d=collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/lw$v/w_counter_collection.csv')):
    d[r['Kernel_Name']].append(float(r['Counter_Value']))
for k,v in d.items():
    if 'pv_synth' in k: print('  WRITE_SIZE KiB', sorted(v)[len(v)//2])
PY
done
