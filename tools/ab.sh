#!/bin/bash
# A/B kernel timing on ONE GPU box: tools/ab.sh TAG_A TAG_B [rounds]   ("base" = the in-tree libnae_gpu.so)
R=${3:-2}
D=$(dirname "$0")/../nodey-audio-editor_amd
for r in $(seq 1 $R); do
  for t in $1 $2; do
    if [ "$t" == "base" ]; then unset NAE_GPU_LIB; else export NAE_GPU_LIB=$D/variants/libnae_gpu_$t.so; fi
    echo "== $t (round $r)"
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-pcie | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  step %.2f ms | ' % d['ms_per_step'] + ' '.join('%s %.2f' % (k.replace('_kernel',''), v['avg_ms']) for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_ms'])[:5]))
"
  done
done
