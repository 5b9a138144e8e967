#!/bin/bash
# round 6: the merged pipeline source (kernels_pvpipe.hip + pv_roles.h) against the previous commit's build (variants/libnae_gpu_prev.so): GPU tests, bits on
# whole graphs, interleaved A/B at 1024 / 512 / 128 / 40 streams
O=gpurun_out/${1:-r6d}; mkdir -p $O
set -o pipefail
python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt; [ $rc -eq 0 ] || exit $rc
V=nodey-audio-editor_amd/variants/libnae_gpu_prev.so
for flow in 1 0 2; do
  NAE_DEBUG=pv_flow=$flow python tools/lib_hash.py > $O/hash_new_$flow.txt 2>&1 || { tail $O/hash_new_$flow.txt; exit 1; }
  NAE_DEBUG=pv_flow=$flow NAE_GPU_LIB=$V python tools/lib_hash.py > $O/hash_prev_$flow.txt 2>&1 || { tail $O/hash_prev_$flow.txt; exit 1; }
  if cmp -s $O/hash_new_$flow.txt $O/hash_prev_$flow.txt; then echo "pv_flow=$flow: merged source = previous build, SAME BITS"; else echo "pv_flow=$flow: DIFFERENT"; diff $O/hash_new_$flow.txt $O/hash_prev_$flow.txt; fi
done | tee $O/bits.txt
for n in 1024 512 128 40; do echo "== $n streams"; N=$n bash tools/ab_libs.sh base prev; done > $O/ab.txt 2>&1; cat $O/ab.txt
