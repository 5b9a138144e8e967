#!/bin/bash
# N-way timing on ONE GPU box: tools/abn.sh ROUNDS tag1 tag2 ...   ("base" = in-tree library)
R=$1; shift
D=$(dirname "$0")/../nodey-audio-editor_amd
for r in $(seq 1 $R); do
  for t in "$@"; do
    if [ "$t" == "base" ]; then unset NAE_GPU_LIB; else export NAE_GPU_LIB=$D/variants/libnae_gpu_$t.so; fi
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-14s step %.2f ms | ' % ('$t', d['ms_per_step']) + ' '.join('%s %.2f' % (k.replace('_kernel',''), v['avg_ms']) for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['avg_ms'])[:4]))
"
  done
done
