#!/bin/bash
# r05: the one-barrier vocoder pipeline (kernels_pvflow.hip) against the two-barrier one (NAE_PV_FLOW=0) on the stream counts a rank of a
# multi-GPU job owns; interleaved rounds on one box.  usage: r05_flow_ab.sh "128 256 384 512" rounds
for r in $(seq 1 ${2:-2}); do
for n in $1; do
for e in "NAE_PV_FLOW=2" "NAE_PV_FLOW=0"; do
  env $e python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt --no-pcie --no-host-path --sustain-seconds 0 --total-streams $n | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
g=lambda n: k.get(n,{}).get('avg_ms',0.0)
print('round $r streams %5d [%-16s] step %7.3f ms | vocoder %6.3f (%s) | spectrum %6.3f | mix %6.3f | clock %.2f GHz | %.3f Mcycles' % ($n, '$e', d['ms_per_step'], g('pv_pipe_kernel')+g('pv_flow_kernel'), 'flow' if g('pv_flow_kernel') else 'pipe', g('spectrum_stereo_kernel'), g('mix_resample_tile_kernel'), d['clock_GHz'], d['ms_per_step']*d['clock_GHz']))
"
done
done
done
