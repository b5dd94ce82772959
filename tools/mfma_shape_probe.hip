// Probe: sustained rate of v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16 in a bare register loop (operands in
// registers, random data), 2 waves per SIMD, every CU busy.  Guidance for the next round's conv3 tile shape
// (MI355X_MICROARCH.md reports ~1.15x for the 16x16x32 form).  Build + run:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2) k32(const half8* in, float* out, int iters) {
    half8 a = in[threadIdx.x], b = in[256 + threadIdx.x];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256, 2) k16(const half8* in, float* out, int iters) {
    half8 a = in[threadIdx.x], b = in[256 + threadIdx.x];
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int blocks = 256 * 2, iters = 20000;
    half8* in;
    float* out;
    hipMalloc(&in, 512 * sizeof(half8));
    hipMalloc(&out, blocks * 256 * sizeof(float));
    _Float16* h = (_Float16*)malloc(512 * 8 * sizeof(_Float16));
    srand(1);
    for (int i = 0; i < 512 * 8; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    hipMemcpy(in, h, 512 * 8 * sizeof(_Float16), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        k32<<<blocks, 256>>>(in, out, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k32<<<blocks, 256>>>(in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        double f32 = (double)blocks * 4 * iters * 4 * (2.0 * 32 * 32 * 16) / (ms * 1e-3) / 1e12;
        k16<<<blocks, 256>>>(in, out, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k16<<<blocks, 256>>>(in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        double f16 = (double)blocks * 4 * iters * 8 * (2.0 * 16 * 16 * 32) / (ms * 1e-3) / 1e12;
        printf("32x32x16: %.0f TFLOP/s   16x16x32: %.0f TFLOP/s   ratio %.3f\n", f32, f16, f16 / f32);
    }
    return 0;
}
