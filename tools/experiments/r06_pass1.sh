#!/bin/bash
# round 6: pass 1 of tiled lone streams on the pipeline's phase-only mode (default) against round 5's one-wave-per-tile kernel (pv_pass1=1): the tiling tests,
# then C3 (one hour of stereo) and other few-long-stream shapes by kernel, both ways
O=gpurun_out/${1:-r6e}; mkdir -p $O
set -o pipefail
python -m pytest tests/test_gpu_stft.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt; [ $rc -eq 0 ] || exit $rc
for env in "" "NAE_DEBUG=pv_pass1=1" "" "NAE_DEBUG=pv_pass1=1"; do echo "== [$env]"; env $env python tools/experiments/r05_c3_kernels.py; done > $O/c3.txt 2>&1; cat $O/c3.txt
for n in 8 40 64; do for env in "" "NAE_DEBUG=pv_pass1=1"; do
  env $env python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-pcie --no-host-path --sustain-seconds 0 --total-streams $n | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('streams %4d  [%-20s] step %.3f ms | ' % ($n, '$env', d['ms_per_step']) + ' '.join('%s %.3f' % (k.replace('_kernel',''), v['avg_ms']) for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_ms'])))
"; done; done > $O/small.txt 2>&1; cat $O/small.txt
