// Box probe: a bare v_mfma_f32_16x16x32_f16 register loop on non-trivial operands, every CU busy at two waves per SIMD.
// bench.py times one launch with HIP events and prints the rate beside its numbers: MI355X devices hold different
// clocks under matrix load (MI355X_MICROARCH.md, DVFS give-back item 5: 12 % apart for a loop without memory traffic),
// so end-to-end figures from two boxes are only comparable next to this figure.  Not on the hot path; no reference
// counterpart.
#include "common.h"

namespace {
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2) mfma_probe_kernel(float* out, int iters, int vary) {
    // four A and four B fragments of pseudo-random halves in (-0.5, 0.5), different per lane and element.  vary = 0: every
    // MFMA multiplies the same pair (what a register loop usually does: the operand buses barely toggle); vary = 1: the
    // pairs rotate, so consecutive instructions see different data on every operand -- what a real kernel does.  The
    // chip holds a lower clock on the second (MI355X_MICROARCH.md, DVFS give-back: zero-filled inputs +19 %).
    half8 a[4], b[4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned ha = (threadIdx.x * 2654435761u + (f * 8 + i) * 40503u) >> 7;
            const unsigned hb = (threadIdx.x * 2246822519u + (f * 8 + i) * 69069u + 12345u) >> 9;
            a[f][i] = (_Float16)((float)(ha & 1023u) * (1.0f / 1024.0f) - 0.5f);
            b[f][i] = (_Float16)((float)(hb & 1023u) * (1.0f / 1024.0f) - 0.5f);
        }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (vary) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc[i], 0, 0, 0);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace

extern "C" int sk_mfma_probe(void* scratch, size_t scratch_bytes, int iters, int vary_operands, double* flops, void* stream) {
    constexpr int kBlocks = 512;   // 256 CUs x 2 workgroups of 4 waves
    SK_CHECK_ARG(scratch != nullptr && scratch_bytes >= (size_t)kBlocks * 256 * sizeof(float),
                 "sk_mfma_probe: scratch must hold %d floats", kBlocks * 256);
    SK_CHECK_ARG(iters > 0, "sk_mfma_probe: iters must be positive");
    mfma_probe_kernel<<<kBlocks, 256, 0, (hipStream_t)stream>>>((float*)scratch, iters, vary_operands);
    SK_CHECK_LAUNCH();
    if (flops) *flops = (double)kBlocks * 4 * iters * 8 * (2.0 * 16 * 16 * 32);
    return SK_OK;
}

// ---- CU-masked streams (round 4 experiment: tools/cu_mask_probe.py, cu_overlap_probe.py, cu_queue_probe.py) ----------------
// The eval step is 75 % MFMA-bound convs that fill every CU's registers and LDS, and 25 % HBM-bound passes that cannot
// co-reside with a conv workgroup.  Two HIP streams with disjoint CU masks let the passes of one tile batch run beside the
// convs of another -- measured, and not adopted (the "passes" are not cheap enough on a small CU share).
// hipExtStreamCreateWithCUMask is the runtime's own API; it lives here so that the stream belongs to the HIP runtime
// instance the kernels are launched from.
namespace {
__global__ void where_kernel(unsigned* out, int spin) {
    // one record per workgroup: XCC id | hardware id (cu / sh / se); a short spin keeps the workgroups resident side by side
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
    long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) {
    }
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
}
}  // namespace

extern "C" int sk_stream_create_cu_mask(const uint32_t* mask_words, int n_words, void** stream_out) {
    SK_CHECK_ARG(mask_words && n_words > 0 && stream_out, "sk_stream_create_cu_mask: bad arguments");
    hipStream_t s = nullptr;
    SK_CHECK_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask_words));
    *stream_out = (void*)s;
    return SK_OK;
}

extern "C" int sk_stream_destroy(void* stream) {
    SK_CHECK_ARG(stream, "sk_stream_destroy: NULL stream");
    SK_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
    return SK_OK;
}

extern "C" int sk_debug_where(unsigned* out, int n_blocks, int spin_cycles, void* stream) {
    SK_CHECK_ARG(out && n_blocks > 0, "sk_debug_where: bad arguments");
    where_kernel<<<n_blocks, 64, 0, (hipStream_t)stream>>>(out, spin_cycles);
    SK_CHECK_LAUNCH();
    return SK_OK;
}
