#!/bin/bash
# round 5: issue priorities of the four roles in the frame-interleaved vocoder pipeline (kG > 1, one workgroup per CU; the 128-stream
# share of an 8-GPU job).  Builds variants/libnae_gpu_prio<tag>.so from patched COPIES of kernels_pvpipe.hip.
#   tools/experiments/r05_kg_prio.sh build      then on the GPU box:   tools/experiments/r05_kg_prio.sh run
R=$(cd "$(dirname "$0")/../.." && pwd)
MAPS="a:0,2,1,1 b:0,1,1,0 c:0,2,2,1 d:0,1,0,1 e:0,3,2,1 f:0,0,0,0 g:1,3,2,2 h:0,2,1,0"
if [ "$1" == "build" ]; then
  for m in $MAPS; do
    tag=${m%%:*}; map=${m##*:}
    mkdir -p /tmp/prio_$tag/csrc
    python3 - "$R" "$tag" "$map" <<'PY'
import sys
r, tag, mp = sys.argv[1:4]
p = [int(x) for x in mp.split(',')]
s = open(r + '/nodey-audio-editor_amd/csrc/kernels_pvpipe.hip').read()
old = 'if (kG > 1 && role == 2) __builtin_amdgcn_s_setprio(1);'
assert old in s
new = 'if (kG > 1) { if (role == 0) __builtin_amdgcn_s_setprio(%d); else if (role == 1) __builtin_amdgcn_s_setprio(%d); else if (role == 2) __builtin_amdgcn_s_setprio(%d); else __builtin_amdgcn_s_setprio(%d); }' % tuple(p)
open('/tmp/prio_%s/csrc/kernels_pvpipe.hip' % tag, 'w').write(s.replace(old, new))
PY
    SRC_PVPIPE=/tmp/prio_$tag/csrc/kernels_pvpipe.hip bash $R/tools/mkvariant.sh prio$tag
  done
else
  for r in 1 2; do
  for m in base $MAPS; do
    tag=${m%%:*}
    if [ "$m" == "base" ]; then unset NAE_GPU_LIB; else export NAE_GPU_LIB=$R/nodey-audio-editor_amd/variants/libnae_gpu_prio$tag.so; fi
    python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt --no-pcie --no-host-path --sustain-seconds 0 --total-streams ${N:-128} | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-12s step %.3f ms | pv_pipe %.3f | clock %.2f' % ('$m', d['ms_per_step'], d['kernels']['pv_pipe_kernel']['avg_ms'], d['clock_GHz']))"
  done
  done
fi
