#!/bin/bash
# build a variant of libnae_gpu.so for A/B timing:  tools/mkvariant.sh TAG [extra stft flags...]
# e.g. tools/mkvariant.sh slp  (re-enables SLP)   tools/mkvariant.sh x -DNAE_FOO=1
set -e
TAG=$1; shift
D=$(dirname "$0")/../nodey-audio-editor_amd
mkdir -p $D/variants
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
STFT="-fno-slp-vectorize"
for a in "$@"; do
  if [ "$a" == "+slp" ]; then STFT=""; else STFT="$STFT $a"; fi
done
for f in nae_api nae_stream nae_swr kernels_nodes nae_wsola; do /opt/rocm/bin/hipcc $COMMON $NODEFLAGS -c $D/csrc/$f.hip -o /tmp/v_$f.o 2>/dev/null & done
/opt/rocm/bin/hipcc $COMMON $STFT -I$D/csrc -c ${SRC_STFT:-$D/csrc/kernels_stft.hip} -o /tmp/v_kernels_stft.o 2>/dev/null &   # SRC_STFT: a patched copy (tools/experiments)
/opt/rocm/bin/hipcc $COMMON $STFT -I$D/csrc -c ${SRC_PVPIPE:-$D/csrc/kernels_pvpipe.hip} -o /tmp/v_kernels_pvpipe.o 2>/dev/null &   # SRC_PVPIPE: a patched copy
/opt/rocm/bin/hipcc $COMMON -fno-slp-vectorize -c $D/csrc/kernels_wsola.hip -o /tmp/v_kernels_wsola.o 2>/dev/null &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/variants/libnae_gpu_$TAG.so /tmp/v_nae_api.o /tmp/v_nae_stream.o /tmp/v_nae_swr.o /tmp/v_kernels_nodes.o /tmp/v_kernels_stft.o /tmp/v_kernels_pvpipe.o /tmp/v_nae_wsola.o /tmp/v_kernels_wsola.o
echo built $D/variants/libnae_gpu_$TAG.so
