#!/bin/bash
# round 6, review item 1: the work-removal levers against the in-tree build (plain f32), on ONE box, interleaved:
#   pk1   = hand-placed packed VOP3P complex arithmetic everywhere (tools/mkvariant.sh pk1 -DNAE_PK=1)
#   pk2   = only the twiddle products packed (-DNAE_PK=2)
#   atan3 = TIMING probe of an atan2 revision 3 (-DNAE_ATAN_PROBE=1: two Newton steps, degree-5 polynomial; other values, so no bit check)
# the FFT alone with clock and watts (tools/ubench/fftpk_bench.hip, both builds), whole graphs bit for bit (pk1, pk2), the C5 step and the
# 128-stream rank share A/B, then every build's step for 6 s with rocm-smi clock and power beside it.
#   gpurun -- bash tools/experiments/r06_pk.sh OUTDIR
O=gpurun_out/${1:-r6a}; mkdir -p $O
set -o pipefail
U=tools/ubench
{ $U/fftpk_bench_0 4 head && $U/fftpk_bench_1 4 && $U/fftpk_bench_0 4 && $U/fftpk_bench_1 4; } > $O/fftpk.md 2>&1 || { cat $O/fftpk.md; exit 1; }
cat $O/fftpk.md
D=nodey-audio-editor_amd/variants
python tools/lib_hash.py > $O/hash_base.txt 2>&1 || { tail -20 $O/hash_base.txt; exit 1; }
for v in pk1 pk2; do
  NAE_GPU_LIB=$D/libnae_gpu_$v.so python tools/lib_hash.py > $O/hash_$v.txt 2>&1 || { tail -20 $O/hash_$v.txt; exit 1; }
  if cmp -s $O/hash_base.txt $O/hash_$v.txt; then echo "$v and plain builds: SAME BITS on $(wc -l < $O/hash_base.txt) graphs"; else echo "$v and plain builds DIFFER"; diff $O/hash_base.txt $O/hash_$v.txt; fi
done | tee $O/bits.txt
bash tools/ab_libs.sh base pk1 pk2 atan3 > $O/ab_c5.txt 2>&1 && cat $O/ab_c5.txt
N=128 bash tools/ab_libs.sh base pk1 pk2 atan3 > $O/ab_128.txt 2>&1 && cat $O/ab_128.txt
for v in base pk1 atan3 base pk1; do
  if [ $v == base ]; then unset NAE_GPU_LIB; else export NAE_GPU_LIB=$D/libnae_gpu_$v.so; fi
  echo "== $v"; python tools/clock_watch.py 6 1024 7 | grep -E "ms per step|settled"
done > $O/watts.txt 2>&1; cat $O/watts.txt
