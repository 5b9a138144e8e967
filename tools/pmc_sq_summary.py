#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/pmc_sq.sh into the committed summaries:

  python tools/pmc_sq_summary.py gpurun_out/<dir> <tag> [sample_frames_per_launch]

writes profiles/<tag>_kernel_stats.csv (verbatim --kernel-trace --stats), profiles/<tag>_sq_stalls.md (issue / stall / LDS
counters per kernel), profiles/<tag>_pmc.md (HBM bytes per launch) and profiles/traffic.json (what bench.py reads for
`roofline.traffic` and `roofline.valu`).

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles; FETCH_SIZE and
WRITE_SIZE are KiB.  gfx950 correction (§HBM): FETCH_SIZE tallies the 128-B requests of a wide coalesced read as 64 B, i.e.
reports half the bytes -> doubled here (calibrated in round 1 on the mix kernel, whose compulsory read is known); WRITE_SIZE is
exact for 16-byte-per-lane streaming stores."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# kernel symbol -> the launch label bench.py reports it under
LABEL = {}      # bench.py labels its launches with the names rocprofv3 prints
WAVES_PER_SIMD = {"pv_pipe_kernel": 8, "spectrum_stereo_kernel": 4, "mix_resample_tile_kernel": 5, "st_td_kernel": 4, "st_aa_kernel": 8}


def compiler_resources():
    """VGPRs / SGPRs / spills / occupancy per kernel as the compiler reports them (-Rpass-analysis=kernel-resource-usage): the
    VGPR_Count field of the rocprofv3 CSV is in allocation granules (half the registers of a wave64 kernel) and its
    LDS_Block_Size misses dynamic LDS, so neither is quoted.  The widest instantiation of a template is listed."""
    import subprocess
    res = {}
    src_dir = os.path.join(ROOT, "nodey-audio-editor_amd", "csrc")
    for tu in ("kernels_stft.hip", "kernels_pvpipe.hip", "kernels_wsola.hip"):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                            "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(src_dir, tu), "-o", "/dev/null"], capture_output=True, text=True)
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = re.sub(r"^_ZN3nae\d+", "", m.group(1))
                name = re.match(r"[a-z0-9_]+", name).group(0)
                cur = {"name": name}
                continue
            for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
                m = re.search(pat, line)
                if m and cur is not None:
                    cur[key] = int(m.group(1))
            if cur is not None and "occ" in cur and "spill" in cur and "vgpr" in cur:
                old = res.get(cur["name"])
                if old is None or cur["occ"] > old["occ"]:        # the instantiation the headline batch runs (8 waves per SIMD for the pipeline)
                    res[cur["name"]] = cur
                cur = None
    return res


def norm(name):
    name = name.split("(")[0].replace("void ", "").replace("nae::", "").strip()
    return re.sub(r"<.*", "", name)


def main():
    d, tag = sys.argv[1], sys.argv[2]
    no_traffic = "--no-traffic" in sys.argv           # a side run (e.g. the WSOLA leg): profiles/traffic.json keeps the headline run's numbers
    args = [a for a in sys.argv[3:] if a != "--no-traffic"]
    sf = int(args[0]) if args else 1024 * 480000
    prof = os.path.join(ROOT, "profiles")
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for path in sorted(glob.glob(os.path.join(d, "p[1-9]*", "**", "*counter_collection.csv"), recursive=True)):
        per_dispatch = collections.defaultdict(float)
        kname = {}
        for r in csv.DictReader(open(path)):
            per_dispatch[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            k = norm(r["Kernel_Name"])
            kname[r["Dispatch_Id"]] = k
            g = lambda *names: next((r[n] for n in names if n in r), "0")
            meta[k] = {"vgpr": int(g("VGPR_Count", "Vgpr_Count")), "lds": int(g("LDS_Block_Size", "Lds_Block_Size")), "wg": int(g("Workgroup_Size"))}
        for (disp, cname), v in per_dispatch.items():
            vals[kname[disp]][cname].append(v)
    med = {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in vals.items()}
    keep = [k for k in med if not k.startswith("__amd") and "fill_uniform" not in k and "clock_probe" not in k]
    keep.sort(key=lambda k: -med[k].get("SQ_WAVE_CYCLES", 0))

    dur = {}
    for path in glob.glob(os.path.join(d, "p0", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(path, os.path.join(prof, f"{tag}_kernel_stats.csv"))
        for r in csv.DictReader(open(path)):
            avg = r.get("AverageNs") or r.get("Average (Nsec)")
            dur[norm(r["Name"])] = float(avg) / 1e6

    out = [f"# {tag}: SQ issue / stall / LDS counters per kernel (median over dispatches; rocprofv3 --pmc, one pass per group: tools/pmc_sq.sh)", "",
           "| counter | " + " | ".join(keep) + " |", "|---|" + "---|" * len(keep)]
    for c in sorted({c for k in keep for c in med[k]}):
        out.append(f"| {c} | " + " | ".join(f"{med[k].get(c, float('nan')):.4g}" for k in keep) + " |")
    out += ["", "Derived (per kernel; avg ms from the --kernel-trace --stats pass):", "",
            "| kernel | avg ms | VGPRs (compiler) / waves per SIMD | waves | wait_any / wave_cycles | wait_inst_any / wave_cycles | active_valu / wave_cycles | VALU instr | trans share | int share | LDS instr | "
            "LDS idx-active cycles per CU | bank-conflict share |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    cres = compiler_resources()
    vg = lambda k: (f"{cres[k]['vgpr']} / {cres[k]['occ']}" + (f" ({cres[k]['spill']} spilled)" if cres[k].get("spill") else "")) if k in cres else "?"
    for k in keep:
        g = lambda c: med[k].get(c, float("nan"))
        wc = g("SQ_WAVE_CYCLES")
        out.append(f"| {k} | {dur.get(k, float('nan')):.3f} | {vg(k)} | {g('SQ_WAVES'):.4g} | {g('SQ_WAIT_ANY') / wc:.3f} | {g('SQ_WAIT_INST_ANY') / wc:.3f} | "
                   f"{g('SQ_ACTIVE_INST_VALU') / wc:.3f} | {g('SQ_INSTS_VALU'):.4g} | {g('SQ_INSTS_VALU_TRANS_F32') / g('SQ_INSTS_VALU'):.3f} | "
                   f"{(g('SQ_INSTS_VALU_INT32') + g('SQ_INSTS_VALU_INT64')) / g('SQ_INSTS_VALU'):.3f} | {g('SQ_INSTS_LDS'):.4g} | {g('SQ_LDS_IDX_ACTIVE') / 256:.4g} | "
                   f"{g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.3f} |")
    open(os.path.join(prof, f"{tag}_sq_stalls.md"), "w").write("\n".join(out) + "\n")

    traffic = {"_source": f"profiles/{tag}_pmc.md, profiles/{tag}_sq_stalls.md (rocprofv3 --pmc passes of this build, tools/pmc_sq.sh)"}
    lines = [f"# {tag}: HBM traffic per launch from rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes)", "", f"sample-frames per launch: {sf}", "",
             "| kernel | avg ms (--kernel-trace --stats) | FETCH_SIZE KiB (raw) | read bytes (x2 gfx950 corr.) | WRITE_SIZE KiB | HBM bytes/launch | B per sample-frame | VGPRs (compiler) / waves per SIMD |",
             "|---|---|---|---|---|---|---|---|"]
    for k in keep:
        if "FETCH_SIZE" not in med[k]:
            continue
        rd, wr = med[k]["FETCH_SIZE"] * 1024 * 2, med[k].get("WRITE_SIZE", 0) * 1024
        traffic[LABEL.get(k, k)] = {"kernel": k, "hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "sample_frames": sf,
                                    "valu_instr_per_launch": med[k].get("SQ_INSTS_VALU"), "lds_idx_active_per_cu": med[k].get("SQ_LDS_IDX_ACTIVE", 0) / 256,
                                    "waves_per_simd": WAVES_PER_SIMD.get(k, 4)}
        lines.append(f"| {k} | {dur.get(k, float('nan')):.3f} | {med[k]['FETCH_SIZE']:.0f} | {rd:.4g} | {med[k].get('WRITE_SIZE', 0):.0f} | {rd + wr:.4g} | {(rd + wr) / sf:.2f} | "
                     f"{vg(k)} |")
    open(os.path.join(prof, f"{tag}_pmc.md"), "w").write("\n".join(lines) + "\n")
    if not no_traffic:
        json.dump(traffic, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
    else:
        json.dump(traffic, open(os.path.join(prof, f"{tag}_traffic.json"), "w"), indent=1)
    print("\n".join(out[-len(keep) - 3:]))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
