mkdir -p gpurun_out/r5b
python -m pytest tests/test_gpu_stft.py -x -q -m gpu > gpurun_out/r5b/tests.txt 2>&1; tail -2 gpurun_out/r5b/tests.txt
V=nodey-audio-editor_amd/variants/libnae_gpu_nocomp5.so
bash tools/ab_env.sh "NAE_SPEC_NARROW=1" "A=1" "NAE_GPU_LIB=$V NAE_SPEC_NARROW=1" "NAE_GPU_LIB=$V" > gpurun_out/r5b/ab.txt 2>&1; cat gpurun_out/r5b/ab.txt
