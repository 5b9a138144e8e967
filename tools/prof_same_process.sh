set -e
# tools/prof_same_process.sh [outdir under gpurun_out, default r4s]: rocprofv3 --kernel-trace --stats around a bench.py run that keeps its
# hipEvent timing on: both clocks from one process
R=${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-r4s}
mkdir -p $R/gpurun_out/$O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$O/p0 -o trace --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-pcie --no-host-path --sustain-seconds 0 > $R/gpurun_out/$O/bench_under_rocprof.json 2> $R/gpurun_out/$O/p0.log
export O
python3 - <<'PY'
import json,glob,csv,os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
O=os.environ["O"]
d=json.load(open(R+"/gpurun_out/"+O+"/bench_under_rocprof.json"))
print("hipEvent (bench.py, same process):", {k:v["avg_ms"] for k,v in d["kernels"].items()}, "ms_per_step", d["ms_per_step"])
for p in glob.glob(R+"/gpurun_out/"+O+"/p0/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n=r["Name"]
        if any(k in n for k in ("pv_pipe","spectrum_stereo","mix_resample")):
            print("rocprofv3:", n[:60], "calls", r.get("Calls"), "avg ms", float(r.get("AverageNs") or r.get("Average (Nsec)"))/1e6)
PY
