mkdir -p gpurun_out/r5f
python -m pytest tests/test_gpu_stft.py tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/r5f/tests.txt 2>&1; tail -2 gpurun_out/r5f/tests.txt
V=nodey-audio-editor_amd/variants/libnae_gpu_prev.so
bash tools/ab_env.sh "NAE_GPU_LIB=$V" "A=1" "NAE_SPEC_NARROW=1" "NAE_SPEC_CHUNK=32" > gpurun_out/r5f/ab.txt 2>&1; cat gpurun_out/r5f/ab.txt
