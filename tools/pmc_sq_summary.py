#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_sq.sh: per kernel, the median over dispatches of every counter
(summed over the chip by rocprofv3), plus the ratios that say what a kernel waits for.

  python tools/pmc_sq_summary.py gpurun_out/<dir> > profiles/rNN_sq_stalls.md

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles."""
import collections
import csv
import glob
import os
import re
import sys


def norm(name):
    name = name.split("(")[0].replace("void ", "").replace("nae::", "").strip()
    return re.sub(r"<.*", "", name)


def main():
    d = sys.argv[1]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sorted(glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True)):
        per_dispatch = collections.defaultdict(float)
        kname = {}
        for r in csv.DictReader(open(path)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            kname[r["Dispatch_Id"]] = norm(r["Kernel_Name"])
        for (disp, cname), v in per_dispatch.items():
            vals[kname[disp]][cname].append(v)
    med = {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in vals.items()}
    keep = [k for k in med if not k.startswith("__amd") and "fill_uniform" not in k]
    keep.sort(key=lambda k: -med[k].get("SQ_WAVE_CYCLES", 0))
    print("# SQ issue / stall counters per kernel (median over dispatches; rocprofv3 --pmc, one pass per group: tools/pmc_sq.sh)\n")
    counters = sorted({c for k in keep for c in med[k]})
    print("| counter | " + " | ".join(keep) + " |")
    print("|---|" + "---|" * len(keep))
    for c in counters:
        print(f"| {c} | " + " | ".join(f"{med[k].get(c, float('nan')):.4g}" for k in keep) + " |")
    print("\nDerived (per kernel):\n")
    print("| kernel | wait_any / wave_cycles | wait_inst_any / wave_cycles | active_inst_any / wave_cycles | active_valu / wave_cycles | active_lds / wave_cycles | "
          "VALU instr | trans share | int32+int64 share | LDS instr | bank-conflict / idx_active | waves |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for k in keep:
        m = med[k]
        g = lambda c: m.get(c, float("nan"))
        wc = g("SQ_WAVE_CYCLES")
        print(f"| {k} | {g('SQ_WAIT_ANY') / wc:.3f} | {g('SQ_WAIT_INST_ANY') / wc:.3f} | {g('SQ_ACTIVE_INST_ANY') / wc:.3f} | {g('SQ_ACTIVE_INST_VALU') / wc:.3f} | "
              f"{g('SQ_ACTIVE_INST_LDS') / wc:.3f} | {g('SQ_INSTS_VALU'):.4g} | {g('SQ_INSTS_VALU_TRANS_F32') / g('SQ_INSTS_VALU'):.3f} | "
              f"{(g('SQ_INSTS_VALU_INT32') + g('SQ_INSTS_VALU_INT64')) / g('SQ_INSTS_VALU'):.3f} | {g('SQ_INSTS_LDS'):.4g} | "
              f"{g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.3f} | {g('SQ_WAVES'):.4g} |")


if __name__ == "__main__":
    main()
