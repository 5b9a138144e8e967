#!/bin/bash
# Profile the bench.py workload on the GPU box: one rocprofv3 run per counter group (--pmc never together with a
# tracing domain other than --kernel-trace), plus one --kernel-trace --stats run.
#   tools/pmc_sq.sh <outdir under gpurun_out> [bench.py args...]
# Then, on either machine:  python tools/pmc_sq_summary.py <outdir> <tag>   (writes profiles/<tag>_*.md/csv, profiles/traffic.json)
set -e
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$OUT
cd /tmp; export TMPDIR=/tmp
# ALT=1: keep the side measurement of the SoundTouch-shaped pitch node in the run (st_td_kernel, st_aa_kernel get counters too)
NOALT="--no-alt"; [ -n "$ALT" ] && NOALT=""
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $NOALT --no-pcie --no-host-path --no-kernel-timing"
rocprofv3 --kernel-trace --stats -d $R/$OUT/p0 -o trace --output-format csv -- $BENCH "$@" > $R/$OUT/p0.log 2>&1
echo "stats pass done"
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
  "SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $R/$OUT/p$i -o pmc --output-format csv -- $BENCH "$@" > $R/$OUT/p$i.log 2>&1
  echo "pmc pass $i done"
done
