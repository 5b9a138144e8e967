#!/bin/bash
# HBM traffic of every kernel of the bench workload (including the SoundTouch-algorithm variant of the pitch node):
# two rocprofv3 passes, FETCH_SIZE and WRITE_SIZE on their own (tools/pmc_sq.sh has the full set for the headline kernels).
#   tools/pmc_traffic.sh <outdir under gpurun_out>   then   python tools/pmc_traffic_summary.py <outdir>
set -e
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$OUT
cd /tmp; export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-host-path --no-kernel-timing"
i=0
for grp in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $R/$OUT/p$i -o pmc --output-format csv -- $BENCH "$@" > $R/$OUT/p$i.log 2>&1
  echo "pmc pass $i done"
done
