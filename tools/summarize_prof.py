#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/...) into the committed summaries under profiles/.

  python tools/summarize_prof.py <tag> <kernel_stats.csv> <fetch counter_collection.csv> <write counter_collection.csv> <sample_frames_per_launch>

Writes profiles/<tag>_kernel_stats.csv (verbatim copy), profiles/<tag>_pmc.md and profiles/traffic.json.
FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE tallies 128-B
requests as 64 B, i.e. reports 1/2 of the bytes of a coalesced streaming read -> doubled here (calibrated on
amix_i2p_kernel, whose compulsory read is known: FETCH_SIZE x 2 = 1.04 x input bytes).  WRITE_SIZE is exact
(fill_uniform_kernel writes 3 932 160 000 B and reports 3 840 000 KiB)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def norm(name):
    """'void nae::pv_synth_kernel<true, 2>(nae::SigViewD, ...)' -> 'pv_synth_kernel'"""
    import re
    name = name.split("(")[0].replace("void ", "").replace("nae::", "").strip()
    name = re.sub(r"<.*", "", name)
    # variants that bench.py / the in-library profiler report under one launch label
    return {"pv_synth2_kernel": "pv_synth_kernel", "spectrum_stereo_kernel": "spectrum_kernel"}.get(name, name)


def per_kernel(path):
    d = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(path)):
        k = norm(r["Kernel_Name"])
        d[k].append(float(r["Counter_Value"]))
        g = lambda *names: next(r[n] for n in names if n in r)       # rocprofv3 csv vs rocpd2csv column names
        meta[k] = {"vgpr": int(g("VGPR_Count", "Vgpr_Count")), "lds": int(g("LDS_Block_Size", "Lds_Block_Size")),
                   "wg": int(g("Workgroup_Size"))}
    return {k: sorted(v)[len(v) // 2] for k, v in d.items()}, meta


def main():
    tag, stats, fetch, write, sf = sys.argv[1:6]
    sf = int(sf)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    shutil.copy(stats, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    f, meta = per_kernel(fetch)
    w, _ = per_kernel(write)
    dur = {}
    for r in csv.DictReader(open(stats)):
        avg = r["AverageNs"] if "AverageNs" in r else r["Average (Nsec)"]
        dur[norm(r["Name"])] = dur.get(norm(r["Name"]), 0.0) + float(avg) / 1e6
    traffic = {}
    lines = [f"# {tag}: HBM traffic per launch from rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes)", "",
             f"sample-frames per launch: {sf}", "",
             "| kernel | avg ms (--kernel-trace --stats) | FETCH_SIZE KiB (raw) | read bytes (x2 gfx950 corr.) | WRITE_SIZE KiB | HBM bytes/launch | B per sample-frame | VGPR | LDS B/WG |",
             "|---|---|---|---|---|---|---|---|---|"]
    for k in sorted(f, key=lambda k: -dur.get(k, 0)):
        if k.startswith("__amd") or k == "fill_uniform_kernel":
            continue
        rd = f[k] * 1024 * 2
        wr = w.get(k, 0) * 1024
        traffic[k] = {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "sample_frames": sf}
        lines.append(f"| {k} | {dur.get(k, float('nan')):.3f} | {f[k]:.0f} | {rd:.4g} | {w.get(k, 0):.0f} | {rd + wr:.4g} | {(rd + wr) / sf:.2f} | {meta[k]['vgpr']} | {meta[k]['lds']} |")
    open(os.path.join(ROOT, "profiles", f"{tag}_pmc.md"), "w").write("\n".join(lines) + "\n")
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
