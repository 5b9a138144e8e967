#!/bin/bash
# builds nodey-audio-editor_amd/variants/libnae_gpu_stamps.so with kernels_pvflow.hip replaced by a copy whose one barrier per step is stamped
# with s_memtime (stamps.inc; the B columns of pipe_stamps.py stay empty).  Diagnostic only.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
D=$HERE/../../nodey-audio-editor_amd
mkdir -p $D/variants
SRC=$D/csrc/kernels_pvflow_stamped.hip
python3 - "$D/csrc/kernels_pvflow.hip" "$HERE/stamps.inc" "$SRC" <<'PY'
import re, sys
src, inc, dst = sys.argv[1:4]
s = open(src).read()
s = re.sub(r"pipe_barrier\(\);", "PIPE_STAMP_BARRIER(t, 0);", s)
s = s.replace("/*pipe:begin*/", "PIPE_TOTAL_BEGIN").replace("/*pipe:r1-end*/", "PIPE_TOTAL_END")
k = s.index("constexpr int kFlowSlots")
s = s[:k] + "} // namespace nae\n" + open(inc).read() + "namespace nae {\n" + s[k:]
open(dst, "w").write(s)
PY
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
/opt/rocm/bin/hipcc $COMMON -fno-slp-vectorize -c $SRC -o /tmp/s_kernels_pvflow.o
rm -f $SRC
OBJS=""
for f in nae_api nae_stream nae_swr kernels_nodes kernels_stft kernels_pvpipe nae_wsola kernels_wsola; do OBJS="$OBJS $D/csrc/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/variants/libnae_gpu_stamps.so $OBJS /tmp/s_kernels_pvflow.o
echo built $D/variants/libnae_gpu_stamps.so
