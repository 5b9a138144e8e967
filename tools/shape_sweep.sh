#!/bin/bash
# the C5 step on the stream counts the ranks of an N-GPU job own (and the counts around nae_pick_pv_shape's thresholds), with the
# vocoder shape chosen by the library and forced: tools/shape_sweep.sh "128 260 384 512 1024" "auto 1 2 4"
# prints ms per step, the vocoder's ms, the in-run clock and cycles per step (= ms x clock): one box, settled clocks (30 steps behind 10)
for n in $1; do
for f in $2; do
  if [ "$f" == "auto" ]; then unset NAE_DEBUG; else export NAE_DEBUG=pv_fps=$f; fi
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt --no-pcie --no-host-path --sustain-seconds 0 --total-streams $n | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
g=lambda n: k.get(n,{}).get('avg_ms',0.0)
pv=g('pv_pipe_kernel')+g('pv_flow_kernel'); p1=g('pv_phase_kernel')+g('pv_scan_kernel')
print('| %5d | %-4s | %7.3f | %6.3f | %6.3f | %6.3f | %6.3f | %.2f | %.3f |' % ($n, '$f', d['ms_per_step'], pv, p1, g('spectrum_stereo_kernel'), g('mix_resample_tile_kernel'), d['clock_GHz'], d['ms_per_step']*d['clock_GHz']))
"
done
done
