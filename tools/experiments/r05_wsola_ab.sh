#!/bin/bash
# round 5: the SoundTouch-shaped pitch node of the bench workload under environment settings (interleaved, one box):
#   tools/experiments/r05_wsola_ab.sh "A=1" "NAE_TD_NC=3" "NAE_TD_NC=5" ["NAE_GPU_LIB=nodey-audio-editor_amd/variants/libnae_gpu_prev.so"]
for r in 1 2 3; do
for e in "$@"; do
  env $e python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie --no-host-path --sustain-seconds 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); a=d['pitch_node_soundtouch_algorithm']
print('[%s] pitch node %.3f ms | ' % ('$e', a['pitch_node_ms']) + ' '.join('%s %.3f' % kv for kv in a['kernels_avg_ms'].items()))"
done
done
