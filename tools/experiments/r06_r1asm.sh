#!/bin/bash
# round 6, review item 1c: frame-interleaved shapes with the finished hop blocks summed and stored by R1 (its barrier interval has slack) instead of R3
# (variants/libnae_gpu_r1asm.so = tools/mkvariant.sh r1asm -DNAE_R1_ASSEMBLE=1) against the shipped build: bits, then ms / clock / cycles at 128, 256 and 64 streams
O=gpurun_out/${1:-r6g}; mkdir -p $O
set -o pipefail
V=nodey-audio-editor_amd/variants/libnae_gpu_r1asm.so
for fps in 2 4; do
  NAE_DEBUG=pv_fps=$fps,pv_flow=0 python tools/lib_hash.py > $O/hash_base_$fps.txt 2>&1 || { tail $O/hash_base_$fps.txt; exit 1; }
  NAE_DEBUG=pv_fps=$fps,pv_flow=0 NAE_GPU_LIB=$V python tools/lib_hash.py > $O/hash_r1_$fps.txt 2>&1 || { tail $O/hash_r1_$fps.txt; exit 1; }
  if cmp -s $O/hash_base_$fps.txt $O/hash_r1_$fps.txt; then echo "pv_fps=$fps: R1 assembly = shipped build, SAME BITS"; else echo "pv_fps=$fps: DIFFERENT"; diff $O/hash_base_$fps.txt $O/hash_r1_$fps.txt; fi
done | tee $O/bits.txt
echo "| streams | build | ms per step | vocoder | pass 1 + 2 | spectrum | mix + transposer | clock GHz | Mcycles per step |" > $O/ab.md
echo "|---|---|---|---|---|---|---|---|---|" >> $O/ab.md
for r in 1 2 3; do for n in 128 256 64; do
  unset NAE_GPU_LIB; bash tools/shape_sweep.sh "$n" "auto" | sed 's/auto/base/' >> $O/ab.md
  export NAE_GPU_LIB=$V; bash tools/shape_sweep.sh "$n" "auto" | sed 's/auto/r1asm/' >> $O/ab.md
done; done; cat $O/ab.md
