#!/bin/bash
# round 5: SQ counters of the wall-clock vector-rate micro-benchmark (one launch per shape): SQ_INSTS_VALU / SQ_BUSY_CYCLES /
# SQ_ACTIVE_INST_VALU per kernel dispatch, to set beside its hipEvent TFLOP/s.   tools/experiments/r05_valu_pmc.sh <outdir under gpurun_out>
set -e
OUT=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $R/$OUT/valu_pmc -o pmc --output-format csv -- $R/tools/ubench/valu_wallclock pmc > $R/$OUT/valu_pmc.md 2> $R/$OUT/valu_pmc.err
ls $R/$OUT/valu_pmc
