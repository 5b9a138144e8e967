#!/bin/bash
# round 5: where a wave of spectrum_stereo_kernel<wide> spends its iteration.  Builds variants/libnae_gpu_specstamps.so from a
# patched COPY of kernels_stft.hip: s_memtime around the sections of the frame loop, summed per wave over its chunk, and written
# over the first 64 bytes of the wave's first output frame (the results are wrong on purpose).  Read with r05_spec_stamps.py.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p /tmp/specstamps/csrc
python3 - "$R" <<'PY'
import sys
r = sys.argv[1]
s = open(r + '/nodey-audio-editor_amd/csrc/kernels_stft.hip').read()
def rep(a, b, count=1):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b, count)
# the kernel template is instantiated twice; the stamps only make sense in the wide path but compile in both
rep('''    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* hann = reinterpret_cast<float*>(smem);
    cf* t1024 = reinterpret_cast<cf*>(smem + NAE_FFT_N * sizeof(float));
    cf* w64 = t1024 + kT1024Pad;
    cf* twa = w64 + 64;
    for (int i = threadIdx.x; i < NAE_FFT_N; i += kThreads) hann[i] = tb.hann[i];''', '''    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned long long t_entry = __builtin_readcyclecounter(), r_entry = __builtin_amdgcn_s_memrealtime();
    float* hann = reinterpret_cast<float*>(smem);
    cf* t1024 = reinterpret_cast<cf*>(smem + NAE_FFT_N * sizeof(float));
    cf* w64 = t1024 + kT1024Pad;
    cf* twa = w64 + 64;
    for (int i = threadIdx.x; i < NAE_FFT_N; i += kThreads) hann[i] = tb.hann[i];''', 2)
rep('''#pragma unroll 1
    for (int f = f0; f < f1; f++) {
        cf v0[8], v1[8];
        u32x4 q[5];''', '''    const unsigned long long t_loop0 = __builtin_readcyclecounter();
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
    auto STAMP = [&](int i) { const unsigned long long t = __builtin_readcyclecounter(); tacc[i] += t - tprev; tprev = t; };
#pragma unroll 1
    for (int f = f0; f < f1; f++) {
        cf v0[8], v1[8];
        u32x4 q[5];''')
rep('''        cf w[8];
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = lds_ld(hw + 64 * j);''', '''        STAMP(0);   // staged reads + stores issued
        cf w[8];
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = lds_ld(hw + 64 * j);''')
rep('''        channel(v0, ma);
        __builtin_amdgcn_sched_barrier(0);      // keep the two channels apart''', '''        __builtin_amdgcn_sched_barrier(0);
        STAMP(1);   // window + loads issued
        channel(v0, ma);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(2);   // channel 0
        __builtin_amdgcn_sched_barrier(0);      // keep the two channels apart''')
rep('''        channel(v1, mb);
        __builtin_amdgcn_sched_barrier(0);
        if (kWide) stage_frame(f);''', '''        channel(v1, mb);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(3);   // channel 1
        if (kWide) stage_frame(f);
        STAMP(4);   // staging writes''')
rep('''        raw[6] = pre[0];
        raw[7] = pre[1];
    }''', '''        raw[6] = pre[0];
        raw[7] = pre[1];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        STAMP(5);   // wait for the loads of frame f + 1 (and, older, the stores of frame f - 1)
    }''')
rep('''        else store_frame(f1 - 1);
    }
}''', '''        else store_frame(f1 - 1);
    }
    if (f1 > f0 && lane < 12) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(obase + (long long)f0 * (2 * NAE_FFT_BINS));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_exit = __builtin_readcyclecounter();
        unsigned long long v = tacc[0];
        for (int i = 1; i < 6; i++) v = lane == i ? tacc[i] : v;
        if (lane == 6) v = t_loop0 - t_entry;          // prologue: table fill, barrier, first frame's loads issued
        if (lane == 7) v = tprev - t_loop0;            // the loop
        if (lane == 8) v = t_exit - tprev;             // epilogue: last frame's stores
        if (lane == 9) v = t_entry;                    // absolute entry time (per-XCD counter)
        if (lane == 10) v = r_entry;                   // 100 MHz wall clock at entry
        if (lane == 11) v = __builtin_amdgcn_s_memrealtime();   // ... at exit
        o[lane] = v;
    }
}''')
open('/tmp/specstamps/csrc/kernels_stft.hip', 'w').write(s)
PY
SRC_STFT=/tmp/specstamps/csrc/kernels_stft.hip bash $R/tools/mkvariant.sh specstamps
