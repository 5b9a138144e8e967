#!/bin/bash
# pass 1 on small batches: pipeline phase-only (A) / one wave per tile with >= 64-frame tiles (B, round 5) / one wave per tile with >= 16-frame tiles (C)
O=gpurun_out/${1:-r6f}; mkdir -p $O
set -o pipefail
python -m pytest tests/test_gpu_stft.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt; [ $rc -eq 0 ] || exit $rc
for n in 4 8 16 40 64; do for env in "NAE_DEBUG=pv_min_ptile=64" "NAE_DEBUG=pv_pass1=1,pv_min_ptile=64" "NAE_DEBUG=pv_pass1=1" "NAE_DEBUG=pv_pass1=1,pv_min_ptile=24" "NAE_DEBUG=pv_min_ptile=16"; do
  env $env python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-pcie --no-host-path --sustain-seconds 0 --total-streams $n | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('streams %4d  [%-36s] step %.3f ms | ' % ($n, '$env', d['ms_per_step']) + ' '.join('%s %.3f' % (k.replace('_kernel',''), v['avg_ms']) for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_ms'])))
"; done; done > $O/small.txt 2>&1; cat $O/small.txt
