export NAE_GPU_LIB=nodey-audio-editor_amd/variants/libnae_gpu_nocomp.so
for c in 32 8 4 2 1; do
  NAE_SPEC_CHUNK=$c python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt --no-pcie --no-host-path --sustain-seconds 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('nocomp chunk $c: spectrum %.3f ms' % d['kernels']['spectrum_stereo_kernel']['avg_ms'])"
done
