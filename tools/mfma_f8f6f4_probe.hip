// Probe for DESIGN.md section 10 (item 3): the block-scaled matrix instruction v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950.
//   1. operand layout check: a 16 x 16 x 128 product of exactly representable e4m3 values against the host, under the
//      hypothesis  lane l, byte j of the 32-byte operand  <->  A[row l & 15][k = 32 (l >> 4) + j],  B[k = 32 (l >> 4) + j][col l & 15]
//      (the K = 32 form's map with four times the K per lane), scales = 2^0; then the E8M0 scale of one K block doubled;
//   2. sustained rates of a bare register loop (8 independent accumulators, 2 waves per SIMD, every CU): f16 16x16x32,
//      fp8 (e4m3) 16x16x128, fp6 (e2m3) 16x16x128.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f8f6f4_probe.hip -o /tmp/f8probe && /tmp/f8probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void one_mfma(const v8i* in, float* out, int scale_a, int scale_b, int lane_lo, int lane_hi) {
    const v8i a = in[threadIdx.x], b = in[64 + threadIdx.x];
    f32x4 c = {0, 0, 0, 0};
    const bool sel = (int)threadIdx.x >= lane_lo && (int)threadIdx.x < lane_hi;
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sel ? scale_a : 0x7f7f7f7f, 0, sel ? scale_b : 0x7f7f7f7f);
    for (int r = 0; r < 4; ++r) out[threadIdx.x * 4 + r] = c[r];   // D[row 4 (l >> 4) + r][col l & 15]
}

template <int FMT>
__global__ void __launch_bounds__(256, 2) loop_f8f6f4(const v8i* in, float* out, int iters) {
    v8i a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], FMT, FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256, 2) loop_f16(const half8* in, float* out, int iters) {
    half8 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static uint8_t e4m3(float v) {   // exactly representable inputs only: +-{0, 0.5, 1, 1.5, 2, 3}
    const uint8_t s = v < 0 ? 0x80 : 0;
    const float a = v < 0 ? -v : v;
    const float tab[] = {0, 0.5f, 1, 1.5f, 2, 3};
    const uint8_t enc[] = {0x00, 0x30, 0x38, 0x3C, 0x40, 0x44};   // bias 7: 1.0 = 0 0111 000
    for (int i = 0; i < 6; ++i)
        if (a == tab[i]) return s | enc[i];
    abort();
}

int main() {
    const float vals[] = {-3, -2, -1.5f, -1, -0.5f, 0, 0.5f, 1, 1.5f, 2, 3};
    static float A[16][128], B[128][16];
    static uint8_t h[128 * 32];
    srand(7);
    for (int m = 0; m < 16; ++m)
        for (int k = 0; k < 128; ++k) A[m][k] = vals[rand() % 11];
    for (int k = 0; k < 128; ++k)
        for (int n = 0; n < 16; ++n) B[k][n] = vals[rand() % 11];
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            h[l * 32 + j] = e4m3(A[l & 15][32 * (l >> 4) + j]);
            h[(64 + l) * 32 + j] = e4m3(B[32 * (l >> 4) + j][l & 15]);
        }
    v8i* din;
    float* dout;
    hipMalloc(&din, sizeof(h));
    hipMalloc(&dout, 512 * 256 * sizeof(float));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    float got[256];
    // E8M0 scale semantics: lanes [lo, hi) pass 2^1 for A (variant 1..4: one K block of 16 lanes each; 5: every lane) or for B (6)
    const int los[] = {0, 0, 16, 32, 48, 0, 0}, his[] = {0, 16, 32, 48, 64, 64, 16};
    for (int variant = 0; variant < 7; ++variant) {
        const bool onb = variant == 6;
        one_mfma<<<1, 64>>>(din, dout, onb ? 0x7f7f7f7f : 0x7f7f7f80, onb ? 0x7f7f7f80 : 0x7f7f7f7f, los[variant], his[variant]);
        hipMemcpy(got, dout, sizeof(got), hipMemcpyDeviceToHost);
        // candidate meanings: (a) lane l scales ITS OWN 32 K elements of its row / column; (b) nothing; report the error under (a)
        double worst = 0, worst_plain = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int m = 4 * (l >> 4) + r, n = l & 15;
                double want = 0, plain = 0;
                for (int k = 0; k < 128; ++k) {
                    const int lane_a = 16 * (k / 32) + m, lane_b = 16 * (k / 32) + n;   // the lanes that hold A[m][k] / B[k][n]
                    const int src = onb ? lane_b : lane_a;
                    const double f = (src >= los[variant] && src < his[variant]) ? 2.0 : 1.0;
                    want += (double)A[m][k] * B[k][n] * f;
                    plain += (double)A[m][k] * B[k][n];
                }
                const double e = want - got[l * 4 + r], e0 = plain - got[l * 4 + r];
                worst = (e < 0 ? -e : e) > worst ? (e < 0 ? -e : e) : worst;
                worst_plain = (e0 < 0 ? -e0 : e0) > worst_plain ? (e0 < 0 ? -e0 : e0) : worst_plain;
            }
        printf("lanes [%2d,%2d) pass scale 2^1 for %s: max |error| under 'a lane scales its own K block' = %g, under 'no effect' = %g\n",
               los[variant], his[variant], onb ? "B" : "A", worst, worst_plain);
    }
    const int blocks = 512, iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        double r[3];
        for (int which = 0; which < 3; ++which) {
            for (int t = 0; t < 2; ++t) {
                hipEventRecord(e0);
                if (which == 0) loop_f16<<<blocks, 256>>>((const half8*)din, dout, iters);
                if (which == 1) loop_f8f6f4<0><<<blocks, 256>>>(din, dout, iters);
                if (which == 2) loop_f8f6f4<2><<<blocks, 256>>>(din, dout, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            hipEventElapsedTime(&ms, e0, e1);
            const double k = which == 0 ? 32 : 128;
            r[which] = (double)blocks * 4 * iters * 8 * (2.0 * 16 * 16 * k) / (ms * 1e-3) / 1e12;
        }
        printf("f16 16x16x32: %.0f TFLOP/s   fp8 16x16x128: %.0f (x%.2f)   fp6 16x16x128: %.0f (x%.2f)\n", r[0], r[1], r[1] / r[0], r[2],
               r[2] / r[0]);
    }
    return 0;
}
