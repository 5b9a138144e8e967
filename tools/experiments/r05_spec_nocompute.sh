#!/bin/bash
# round 5: the stereo spectrum kernel WITHOUT its FFT / split / magnitudes (wrong results on purpose): loads, window and the
# frame's stores only — what the byte movement alone costs with dword pieces (NAE_SPEC_NARROW=1) and with 16-byte pieces.
# Builds nodey-audio-editor_amd/variants/libnae_gpu_nocomp5.so from a patched COPY of kernels_stft.hip.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p /tmp/nocomp5/csrc
python3 - "$R" <<'PY'
import sys
r = sys.argv[1]
s = open(r + '/nodey-audio-editor_amd/csrc/kernels_stft.hip').read()
a = '        channel(v0, ma);\n'
b = '        channel(v1, mb);\n'
assert a in s and b in s
s = s.replace(a, '#pragma unroll\n        for (int i = 0; i < 8; i++) ma[i] = v0[i].x + v0[i].y;\n        ma[8] = v0[0].x;\n')
s = s.replace(b, '#pragma unroll\n        for (int i = 0; i < 8; i++) mb[i] = v1[i].x + v1[i].y;\n        mb[8] = v1[0].x;\n')
s = s.replace('#include "stft_common.h"', '#include "stft_common.h"  // (patched copy: no compute)')
open('/tmp/nocomp5/csrc/kernels_stft.hip', 'w').write(s)
PY
SRC_STFT=/tmp/nocomp5/csrc/kernels_stft.hip bash $R/tools/mkvariant.sh nocomp5
