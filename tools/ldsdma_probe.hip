// How fast can a CU fill LDS?  Every wave of every workgroup loops over a small (L2-resident) buffer and moves 1 KiB per
// instruction into LDS, either with LDS-DMA (buffer_load_dwordx4 ... lds) or through registers (global_load_dwordx4 +
// ds_write_b128).  Prints bytes per clock and CU for 1, 2, 4, 8 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/ldsdma_probe.hip -o gpurun_out/ldsdma_probe && gpurun_out/ldsdma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ void __launch_bounds__(512) fill_kernel(const char* src, int src_kib, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    char* my = lds + w * DEPTH * 1024;
    const char* base = src + (size_t)((blockIdx.x * 8 + w) % src_kib) * 1024 + lane * 16;
    f4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + ((it * DEPTH + d) % 64) * 1024),
                                                 (__attribute__((address_space(3))) void*)(my + d * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            f4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = *(const f4*)(base + ((it * DEPTH + d) % 64) * 1024);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) *(f4*)(my + d * 1024 + lane * 16) = v[d];
        }
    }
    __syncthreads();
    acc = *(f4*)(my + lane * 16);
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

template <int MODE, int DEPTH>
double run(const char* src, int src_kib, float* sink, int waves, int iters) {
    hipFuncSetAttribute((const void*)fill_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * DEPTH * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // one workgroup per CU: dynamic LDS sized so that only one fits
    const size_t lds = 100 * 1024;
    hipFuncSetAttribute((const void*)fill_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    fill_kernel<MODE, DEPTH><<<256, waves * 64, lds>>>(src, src_kib, 10, sink);
    hipEventRecord(e0);
    fill_kernel<MODE, DEPTH><<<256, waves * 64, lds>>>(src, src_kib, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)waves * DEPTH * 1024.0 * iters / (ms * 1e-3);   // bytes per second per CU
}

int main() {
    const int src_kib = 2048;   // 2 MiB: L2 resident
    char* src;
    float* sink;
    hipMalloc(&src, (size_t)(src_kib + 64) * 1024);
    hipMemset(src, 0, (size_t)(src_kib + 64) * 1024);
    hipMalloc(&sink, 16);
    const int iters = 2000;
    for (int waves : {1, 2, 4, 8}) {
        double a = run<0, 4>(src, src_kib, sink, waves, iters), b = run<0, 8>(src, src_kib, sink, waves, iters);
        double c = run<1, 4>(src, src_kib, sink, waves, iters), d = run<1, 8>(src, src_kib, sink, waves, iters);
        printf("waves/CU %d: LDS-DMA depth4 %.1f GB/s/CU depth8 %.1f | via registers depth4 %.1f depth8 %.1f  (x256 CUs: %.2f / %.2f TB/s)\n", waves,
               a / 1e9, b / 1e9, c / 1e9, d / 1e9, b * 256 / 1e12, d * 256 / 1e12);
    }
    return 0;
}
