#!/bin/bash
# round 5 probes (WRONG results on purpose) of the frame-interleaved vocoder's light barrier interval, from patched copies of kernels_pvpipe.hip:
#   kgprobe1: R3 assembles the previous step's block but does not store it     kgprobe2: neither assembles nor stores
R=$(cd "$(dirname "$0")/../.." && pwd)
for v in 1 2; do
mkdir -p /tmp/kgprobe$v/csrc
python3 - "$R" "$v" <<'PY'
import sys
r, v = sys.argv[1], sys.argv[2]
s = open(r + '/nodey-audio-editor_amd/csrc/kernels_pvpipe.hip').read()
if v == '1':
    old = "                store_block(fz - kG - 3, o);"
    assert s.count(old) == 1
    s = s.replace(old, "                if (p.mid_len < 0) store_block(fz - kG - 3, o); else asm volatile(\"\" :: \"v\"(o[0]), \"v\"(o[1]), \"v\"(o[2]), \"v\"(o[3]));")
else:
    old = "            const bool finish = kG > 1 && had && fz - kG - 3 >= b0;"
    assert s.count(old) == 1
    s = s.replace(old, "            const bool finish = kG > 1 && had && fz - kG - 3 >= b0 && p.mid_len < 0;")
open('/tmp/kgprobe%s/csrc/kernels_pvpipe.hip' % v, 'w').write(s)
PY
SRC_PVPIPE=/tmp/kgprobe$v/csrc/kernels_pvpipe.hip bash $R/tools/mkvariant.sh kgprobe$v
done
