export NAE_GPU_LIB=nodey-audio-editor_amd/variants/libnae_gpu_rich.so
for r in 1 2; do
for cfg in "" "NAE_SPEC_RICH=1" "NAE_SPEC_RICH=1 NAE_SPEC_CHUNK=16" "NAE_SPEC_RICH=1 NAE_SPEC_CHUNK=8" "NAE_SPEC_RICH=1 NAE_SPEC_CHUNK=4" "NAE_SPEC_RICH=1 NAE_SPEC_CHUNK=64"; do
  env $cfg python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt --no-pcie --no-host-path --sustain-seconds 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$cfg] spectrum %.3f ms  step %.3f' % (d['kernels']['spectrum_stereo_kernel']['avg_ms'], d['ms_per_step']))"
done
done
