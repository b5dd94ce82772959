// Two questions about filling LDS on gfx950, asked of the hardware:
//  (1) how fast can a CU fill LDS from an L2-resident buffer with LDS-DMA (buffer/global_load ... lds, 1 KiB per
//      instruction) -- 1, 2, 4, 8 waves per CU, 4 or 8 instructions in flight per wave;
//  (2) what does ONE such instruction cost a wave that is streaming MFMAs: a loop of 32 independent
//      v_mfma_f32_32x32x16_f16 per iteration (one wave per SIMD, or two) with N LDS-DMA instructions, or N plain
//      global_load_dwordx4 into registers (+ ds_write_b128 of the previous iteration's data), placed between them.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/ldsdma_probe.hip -o /tmp/ldsdma_probe && /tmp/ldsdma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int DEPTH>
__global__ void __launch_bounds__(512) fill_kernel(const char* src, int src_kib, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    char* my = lds + w * DEPTH * 1024;
    const char* base = src + (size_t)((blockIdx.x * 8 + w) % src_kib) * 1024 + lane * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + ((it * DEPTH + d) % 64) * 1024),
                                             (__attribute__((address_space(3))) void*)(my + d * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const f4 acc = *(f4*)(my + lane * 16);
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

// MODE 0: MFMAs only; 1: + N LDS-DMA per iteration; 2: + N loads into registers, stored to LDS one iteration later
// BURST: the N instructions back to back at the top of the iteration instead of one every fourth MFMA
template <int MODE, int N, bool BURST = false>
__global__ void __launch_bounds__(512) mfma_kernel(const char* src, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    char* my = lds + w * 8 * 1024;
    const char* base = src + (size_t)(blockIdx.x * 8 + w) * 1024 + lane * 16;
    f16v acc[8];
    for (int k = 0; k < 8; ++k)
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
    h8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)(0.001f * (lane + j));
        b[j] = (_Float16)(0.002f * (lane - j));
    }
    f4 st[N > 0 ? N : 1];
    for (int d = 0; d < (N > 0 ? N : 1); ++d) st[d] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (BURST ? (m == 0) : (m % 4 == 1 && m / 4 < N))
#pragma unroll
              for (int d = BURST ? 0 : m / 4; d < (BURST ? N : m / 4 + 1); ++d) {
                if (MODE == 1) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + ((it * 8 + d) % 64) * 1024 * 8),
                                                     (__attribute__((address_space(3))) void*)(my + d * 1024), 16, 0, 0);
                } else if (MODE == 2) {
                    *(f4*)(my + d * 1024 + lane * 16) = st[d];   // the previous iteration's load (the compiler waits for it)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(st[d]) : "v"(base + ((it * 8 + d) % 64) * 1024 * 8) : "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
              }
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0.0f;
    for (int k = 0; k < 8; ++k) t += acc[k][0];
    for (int d = 0; d < (N > 0 ? N : 1); ++d) t += st[d][0];
    __syncthreads();
    t += *(float*)(my + lane * 4);
    if (t == 123.456f) sink[0] = t;
}

template <typename K>
static float timed(K launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch(20);
    hipEventRecord(e0);
    launch(2000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int MODE, int N, bool BURST = false>
static float mfma_ms(const char* src, float* sink, int waves) {
    const size_t lds = 100 * 1024;   // one workgroup per CU
    hipFuncSetAttribute((const void*)mfma_kernel<MODE, N, BURST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return timed([&](int iters) { mfma_kernel<MODE, N, BURST><<<256, waves * 64, lds>>>(src, iters, sink); });
}

template <int DEPTH>
static double fill_rate(const char* src, float* sink, int waves) {
    const size_t lds = 100 * 1024;
    hipFuncSetAttribute((const void*)fill_kernel<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const float ms = timed([&](int iters) { fill_kernel<DEPTH><<<256, waves * 64, lds>>>(src, 2048, iters, sink); });
    return (double)waves * DEPTH * 1024.0 * 2000 / (ms * 1e-3);
}

int main() {
    char* src;
    float* sink;
    hipMalloc(&src, (size_t)8 << 20);
    hipMemset(src, 0, (size_t)8 << 20);
    hipMalloc(&sink, 16);
    for (int waves : {1, 2, 4, 8})
        printf("fill: %d waves/CU  LDS-DMA 4 in flight %.1f GB/s/CU, 8 in flight %.1f GB/s/CU\n", waves, fill_rate<4>(src, sink, waves) / 1e9,
               fill_rate<8>(src, sink, waves) / 1e9);
    for (int waves : {4, 8}) {
        const float base = mfma_ms<0, 0>(src, sink, waves);
        const double cyc = 2.4e6 / 2000.0;   // ms -> cycles per iteration at 2.4 GHz (nominal)
        const double per_wave_iters = waves / 4.0;   // waves sharing a SIMD
        printf("mfma stream, %d wave(s)/SIMD: %.0f cycles per 32 MFMAs and wave (nominal 2.4 GHz)\n", waves / 4, base * cyc / per_wave_iters);
        const float d2 = mfma_ms<1, 2>(src, sink, waves), d8 = mfma_ms<1, 8>(src, sink, waves);
        const float r2 = mfma_ms<2, 2>(src, sink, waves), r8 = mfma_ms<2, 8>(src, sink, waves);
        printf("   + LDS-DMA:            2 per iteration %+.0f cycles each, 8 per iteration %+.0f cycles each\n", (d2 - base) * cyc / per_wave_iters / 2,
               (d8 - base) * cyc / per_wave_iters / 8);
        printf("   + load + ds_write:    2 per iteration %+.0f cycles each, 8 per iteration %+.0f cycles each\n", (r2 - base) * cyc / per_wave_iters / 2,
               (r8 - base) * cyc / per_wave_iters / 8);
        const float b8 = mfma_ms<1, 8, true>(src, sink, waves), c8 = mfma_ms<2, 8, true>(src, sink, waves);
        printf("   back to back (8):     LDS-DMA %+.0f cycles each, load + ds_write %+.0f cycles each\n", (b8 - base) * cyc / per_wave_iters / 8,
               (c8 - base) * cyc / per_wave_iters / 8);
    }
    return 0;
}
