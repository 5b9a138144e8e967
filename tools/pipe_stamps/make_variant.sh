#!/bin/bash
# builds nodey-audio-editor_amd/variants/libnae_gpu_stamps.so: the product sources, with kernels_pvpipe.hip replaced by a copy in
# which every marked barrier ( /*A*/, /*B*/ ) is stamped with s_memtime (tools/pipe_stamps/stamps.inc).  Diagnostic only.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
D=$HERE/../../nodey-audio-editor_amd
mkdir -p $D/variants
SRC=$D/csrc/kernels_pvpipe_stamped.hip
python3 - "$D/csrc/kernels_pvpipe.hip" "$HERE/stamps.inc" "$SRC" <<'PY'
import re, sys
src, inc, dst = sys.argv[1:4]
s = open(src).read()
s = re.sub(r"pipe_barrier\(\);\s*/\*A\*/", "PIPE_STAMP_BARRIER(t, 0);", s)
s = re.sub(r"pipe_barrier\(\);\s*/\*B\*/", "PIPE_STAMP_BARRIER(t, 1);", s)
s = s.replace("/*pipe:begin*/", "PIPE_TOTAL_BEGIN").replace("/*pipe:r1-end*/", "PIPE_TOTAL_END")
# the scaffolding goes behind pipe_barrier()'s definition
k = s.index("// Issue priority")
s = s[:k] + "} // namespace nae\n" + open(inc).read() + "namespace nae {\n" + s[k:]
open(dst, "w").write(s)
PY
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
for f in nae_api nae_stream nae_swr kernels_nodes nae_wsola; do /opt/rocm/bin/hipcc $COMMON -c $D/csrc/$f.hip -o /tmp/s_$f.o 2>/dev/null & done
/opt/rocm/bin/hipcc $COMMON -fno-slp-vectorize -c $D/csrc/kernels_stft.hip -o /tmp/s_kernels_stft.o 2>/dev/null &
/opt/rocm/bin/hipcc $COMMON -fno-slp-vectorize -c $SRC -o /tmp/s_kernels_pvpipe.o 2>/dev/null &
/opt/rocm/bin/hipcc $COMMON -fno-slp-vectorize -c $D/csrc/kernels_wsola.hip -o /tmp/s_kernels_wsola.o 2>/dev/null &
wait
rm -f $SRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/variants/libnae_gpu_stamps.so /tmp/s_nae_api.o /tmp/s_nae_stream.o /tmp/s_nae_swr.o /tmp/s_kernels_nodes.o /tmp/s_kernels_stft.o /tmp/s_kernels_pvpipe.o /tmp/s_nae_wsola.o /tmp/s_kernels_wsola.o
echo built $D/variants/libnae_gpu_stamps.so
