// Shared helpers for libskoots_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <cmath>
#include <stdio.h>

#include "../../include/skoots_hip.h"

// 16-bit floating type of the matrix-core operands and of every 16-bit activation / gradient tensor.  The library is
// fp16 (the inference dtype BASELINE's configs name, fp16 autocast at skoots/lib/eval.py:142).  The sources on the
// training step's path (conv3d.hip, unet_misc.hip, train.hip) are compiled a second time with -DSK_BF16 -- the same
// kernels with bf16 storage and v_mfma_*_bf16, the dtype of the reference's training step (train/engine.py:68,107-109)
// -- and the entry points of that build carry the suffix _bf16 (Makefile: llvm-objcopy --redefine-syms).
#ifdef SK_BF16
typedef __bf16 t16;
#define SK_MFMA_32x32x16_T16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define SK_MFMA_16x16x32_T16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define SK_DS_READ_TR16_B64 __builtin_amdgcn_ds_read_tr16_b64_v4bf16
#define SK_TR16_ELEM __bf16
#define SK_DOT2_T16(a, b, c) __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false)
#else
typedef _Float16 t16;
#define SK_MFMA_32x32x16_T16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define SK_MFMA_16x16x32_T16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define SK_DS_READ_TR16_B64 __builtin_amdgcn_ds_read_tr16_b64_v4f16
#define SK_TR16_ELEM __fp16
#define SK_DOT2_T16(a, b, c) __builtin_amdgcn_fdot2(a, b, c, false)
#endif
typedef t16 t16x2 __attribute__((ext_vector_type(2)));

namespace sk {

void set_error(const char* fmt, ...);

// -DSK_TIMING builds only: where the conv kernels dump their per-wave phase cycle sums (sk_debug_set_timing_buffer).
// One record layout for every kernel: [4096 workgroups][4 waves][kTimingSlots] int64; a launch whose buffer is absent or
// smaller than that does not dump.
constexpr int kTimingSlots = 16;
constexpr int kTimingBlocks = 4096;
long long* timing_buffer();

#define SK_CHECK_ARG(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            ::sk::set_error(__VA_ARGS__);  \
            return SK_ERR_ARG;             \
        }                                  \
    } while (0)

#define SK_CHECK_HIP(expr)                                                            \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            ::sk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                            __FILE__, __LINE__);                                      \
            return SK_ERR_HIP;                                                        \
        }                                                                             \
    } while (0)

#define SK_CHECK_LAUNCH()                                                             \
    do {                                                                              \
        hipError_t _e = hipGetLastError();                                            \
        if (_e != hipSuccess) {                                                       \
            ::sk::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),\
                            __FILE__, __LINE__);                                      \
            return SK_ERR_HIP;                                                        \
        }                                                                             \
    } while (0)

static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// HOST: fp32 -> OCP e4m3fn code (round to nearest even, saturating at 448; the device's v_cvt_pk_fp8_f32 on gfx950)
static inline uint8_t f32_to_e4m3(float v) {
    const uint8_t sgn = std::signbit(v) ? 0x80 : 0x00;
    float a = std::fabs(v);
    if (!(a == a)) return 0x7F;
    if (a >= 448.0f) return sgn | 0x7E;
    if (a < 0.0009765625f) return sgn;                              // below half of the smallest subnormal 2^-9
    int e;
    (void)std::frexp(a, &e);                                        // a = m * 2^e, m in [0.5, 1)
    e -= 1;                                                         // a in [2^e, 2^(e+1))
    if (e < -6) {                                                   // subnormal: units of 2^-9
        const int q = (int)std::nearbyint(a * 512.0f);
        return sgn | (uint8_t)(q >= 8 ? 0x08 : q);
    }
    int m = (int)std::nearbyint((a / std::ldexp(1.0f, e) - 1.0f) * 8.0f);
    if (m == 8) {
        m = 0;
        ++e;
    }
    if (e > 8) return sgn | 0x7E;
    const int code = ((e + 7) << 3) | m;
    return sgn | (uint8_t)(code > 0x7E ? 0x7E : code);
}

// Memory-bound grids: cap and grid-stride (256 CUs x 8 blocks).
static inline unsigned stream_grid(int64_t n, int block, int per_thread = 1) {
    int64_t g = (n + (int64_t)block * per_thread - 1) / ((int64_t)block * per_thread);
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return (unsigned)g;
}

#ifdef __HIPCC__
// fp32 -> 16-bit of a value that is itself the fp32 result of a multiply: the empty asm keeps the product an fp32 value
// of its own, so the conversion is a plain v_cvt (two roundings, what gn_silu_kernel and the torch oracle do).  Without
// it the compiler may fuse multiply + conversion into v_fma_mixlo_f16 -- ONE rounding, a different last bit in about
// one value of two thousand -- and does so in some kernels and not in others (it did inside the conv kernels' in-LDS
// activation, not in the GroupNorm pass: tests/test_hip_unet.py::test_conv_activates_a_raw_source_in_lds).
// The "zero position" of a staged plane, read by a tap that leaves the tile through a z face.  SK_ZERO_WINDOW = 1 (default,
// round 4): a 256-byte window of zeros (4 positions) and the tap reads `zero + (its own address & 255)` -- the banks its
// data read would have used, so a ds_read_b128 lane group stays conflict-free (conv3_px_kernel's finding, round 3:
// with ONE shared 64-byte zero line the four lanes of a z-face voxel each land on another lane's banks, 22-35 % of the
// LDS-active cycles in conflicts).  SK_ZERO_WINDOW = 0 builds the one-line form for the A/B (tools/kernel_ab.sh).
#ifndef SK_ZERO_WINDOW
#define SK_ZERO_WINDOW 1
#endif
constexpr int kZeroPos = SK_ZERO_WINDOW ? 4 : 1;   // zero positions behind the nposp staged ones of every plane slot
__device__ inline int zero_of(int zero_addr, int addr) { return SK_ZERO_WINDOW ? zero_addr + (addr & 255) : zero_addr; }

__device__ inline t16 round_t16(float v) {
    asm("" : "+v"(v));
    return (t16)v;
}

// Buffer loads: the hardware bounds check returns 0 for an offset past the descriptor's byte count, so a
// masked lane passes kOob instead of branching around the load (a branch per load lets the compiler chain
// load -> wait -> use, one memory latency at a time).  The descriptor must be wave-uniform.
constexpr unsigned kOob = 0xFFFFFFF0u;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    // readfirstlane makes the uniformity provable: otherwise every load becomes a waterfall loop
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* q = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ inline float buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ inline f32x4_t buf_load_f32x4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
#endif

}  // namespace sk
