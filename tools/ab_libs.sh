#!/bin/bash
# interleaved A/B of library variants on one box: ab_libs.sh TAG... ("base" = in-tree)
D=$(dirname "$0")/../nodey-audio-editor_amd
for r in 1 2 3; do
for t in "$@"; do
  if [ "$t" == "base" ]; then unset NAE_GPU_LIB; else export NAE_GPU_LIB=$D/variants/libnae_gpu_$t.so; fi
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt --no-pcie --no-host-path --sustain-seconds 0 --total-streams ${N:-1024} | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-8s step %.3f ms | ' % ('$t', d['ms_per_step']) + ' '.join('%s %.3f' % (k.replace('_kernel',''), v['avg_ms']) for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_ms'])[:3]))
"
done
done
