#!/usr/bin/env python3
"""Instruction-mix summary of the gfx950 ISA of one HIP translation unit:  python tools/isa_stats.py csrc/kernels_stft.hip"""
import collections
import os
import re
import subprocess
import sys

src = sys.argv[1]
out = "/tmp/isa_%d.s" % os.getpid()
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
labels = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", s, re.M) if "kernel" in m.group(1)]
for i, (pos, name) in enumerate(labels):
    end = s.find("s_endpgm", pos)
    # take up to the LAST s_endpgm before the next label / metadata
    nxt = labels[i + 1][0] if i + 1 < len(labels) else s.find(".amdgpu_metadata", pos)
    body = s[pos:nxt]
    ops = collections.Counter(l.split()[0] for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";")))
    grp = collections.Counter()
    for k, v in ops.items():
        g = ("valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_")
             else "vmem" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
        grp[g] += v
    short = re.sub(r"^_ZN3nae\d+", "", name)[:28]
    print(f"{short:30s} total {sum(ops.values()):5d} {dict(grp)}")
    print("      ", ops.most_common(18))
for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", s, re.S):
    print(re.sub(r"^_ZN3nae\d+", "", m.group(2))[:28], "agpr", m.group(1), "sgpr", m.group(3), "vgpr", m.group(4), "spill", m.group(5))
os.remove(out)
