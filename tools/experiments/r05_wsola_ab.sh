#!/bin/bash
# round 5: the SoundTouch-shaped pitch node of the bench workload, the in-tree build against variants/libnae_gpu_prev.so (interleaved, one box)
for r in 1 2 3; do
for t in base prev; do
  if [ "$t" == "base" ]; then unset NAE_GPU_LIB; else export NAE_GPU_LIB=nodey-audio-editor_amd/variants/libnae_gpu_$t.so; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-host-path --sustain-seconds 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); a=d['pitch_node_soundtouch_algorithm']
print('%-5s pitch node %.3f ms | ' % ('$t', a['pitch_node_ms']) + ' '.join('%s %.3f' % kv for kv in a['kernels_avg_ms'].items()))"
done
done
