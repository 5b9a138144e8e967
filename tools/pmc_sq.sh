#!/bin/bash
# SQ issue / stall counters of the C5 graph kernels, one rocprofv3 --pmc pass per counter group (no tracing domains
# besides --kernel-trace).  Usage on the GPU box:  tools/pmc_sq.sh <outdir> [bench.py args...]
# Then: python tools/pmc_sq_summary.py <outdir> > profiles/rNN_sq_stalls.md
set -e
OUT=$1; shift
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVES" \
  "GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $R/$OUT/p$i -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-kernel-timing "$@" > $R/$OUT/p$i.log 2>&1
  echo "pass $i done"
done
