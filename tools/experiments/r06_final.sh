#!/bin/bash
# round 6: the round's evidence from the SHIPPED build, one box: GPU tests, the tiny-input fuzz, rocprofv3 passes (stats + counters), both clocks in one
# process, the default bench line, the C5 step for 8 s with rocm-smi clock and power (joules per step), every BASELINE config, the strong-scaling proxy
O=gpurun_out/${1:-r6p}; mkdir -p $O
set -o pipefail
python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt; [ $rc -eq 0 ] || exit $rc
python tests/tools/fuzz_tiny.py > $O/fuzz_tiny.txt 2>&1; rc=$?; tail -2 $O/fuzz_tiny.txt; [ $rc -eq 0 ] || exit $rc
bash tools/pmc_sq.sh $O/pmc > $O/pmc.log 2>&1; rc=$?; tail -2 $O/pmc.log; [ $rc -eq 0 ] || exit $rc
bash tools/prof_same_process.sh ${1:-r6p}/v2 > $O/same_process.txt 2>&1; rc=$?; cat $O/same_process.txt; [ $rc -eq 0 ] || exit $rc
python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; tail -2 $O/bench.err; [ $rc -eq 0 ] || exit $rc
python tools/clock_watch.py 8 1024 7 2>&1 | grep -E "ms per step|settled" | tee $O/watts.txt
python tests/tools/bench_configs.py > $O/configs.md 2> $O/configs.err; rc=$?; cat $O/configs.md; [ $rc -eq 0 ] || { tail -5 $O/configs.err; exit $rc; }
echo "| streams | shape | ms per step | vocoder | pass 1 + 2 | spectrum | mix + transposer | clock GHz | Mcycles per step |" > $O/shape_sweep.md
echo "|---|---|---|---|---|---|---|---|---|" >> $O/shape_sweep.md
bash tools/shape_sweep.sh "1024 512 256 128 64 16" "auto" >> $O/shape_sweep.md 2>&1; cat $O/shape_sweep.md
