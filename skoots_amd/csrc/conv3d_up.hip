// Decoder 3x3x3 convolution over torch.cat([skip, interpolate(x, scale 2, nearest)]) with the nearest-neighbour
// upsample FOLDED INTO THE WEIGHTS of the upsampled half -- same result as sk_conv3d with an upsampled second source,
// 35 % fewer matrix instructions.
//
// Replaces the first conv of each decoder level of the network that skoots/lib/utils.py:17-107 builds (graph:
// oracle/unet_spec.py) and that eval runs at skoots/lib/eval.py:142-143.
//
// Why it folds.  U[x] = L[x >> 1].  Along one axis an output voxel of parity p reads U at x-1, x, x+1, i.e.
//   p = 0 (x = 2m):     L[m-1], L[m], L[m]     ->  w[-1] L[m-1] + (w[0] + w[+1]) L[m]
//   p = 1 (x = 2m + 1): L[m], L[m], L[m+1]     ->  (w[-1] + w[0]) L[m] + w[+1] L[m+1]
// so the 27 taps on U collapse to 2 x 2 x 2 = 8 taps on L with weights that depend on the output voxel's parity class
// (px, py, pz): 8 tap-chunks instead of 27 for the upsampled channels (zero padding of U at the tile faces = zero
// padding of L: the folded sums only ever pair taps that read the same L voxel).  The sums are formed on the host in
// fp32 and rounded to fp16 once (sk_conv3d_pack_weight_upfold_host).
//
// What it costs: the A operand (weights) of an MFMA must be the same for all its voxel columns, so the columns of a
// matrix tile must share (py, pz) and an output plane has one px.  Hence a different voxel -> column map than
// conv3d.hip's: a workgroup owns K whole rows of the LOW-resolution (y, z) plane (K * Zl <= 32 positions) = the
// 2K x Zt fine voxels above them; wave w = parity class (py, pz) = (w >> 1, w & 1) owns the <= 32 fine voxels of that
// class, column m <-> low-resolution position m of the segment.  The staged fine planes are stored DE-INTERLEAVED --
// four sub-planes (y parity, z parity), each linear over (y >> 1, z >> 1) with no z halo -- so that for every tap the 16
// columns of an MFMA read 16 CONSECUTIVE positions (+ a wave-uniform offset): the conflict-free swizzle of
// conv3_m16_kernel carries over unchanged.  z faces: a lane whose tap would wrap into the neighbouring row reads the
// plane's zero position (conv3d.hip's linear mode).  The low-resolution source is staged as it is (4 planes of
// (K + 2) rows) in the ring slots of the fine planes the step is done with; fine planes XS, XS + 1 survive the upsampled
// phases and are planes 0, 1 of the next step (one skip chunk).
//
// 32 output channels per launch on v_mfma_f32_16x16x32_f16 (COUT 64: two launches); B fragments in half-row bodies one
// body ahead, weight rows one (skip chunks; rows 0, 1 from LDS when there is one skip chunk) or two (upsampled chunks)
// rows ahead; everything else (x-marching ring, LDS-DMA through buffer descriptors, permlane epilogue, GroupNorm partials on
// v_dot2c) as in conv3_m16_kernel, whose comments explain those parts.
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <vector>

#include "common.h"

namespace {

typedef t16 half8 __attribute__((ext_vector_type(8)));
typedef t16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPosBytes = 64;    // 32 channels x fp16
constexpr int kMaxDma = 3;       // LDS-DMA wave-instructions per fine plane per wave (nposp <= 192)
constexpr int kMaxDmaL = 2;      // per low-resolution plane per wave (nposl <= 128)
constexpr int kSkipFrags = 54;   // fragments of a skip chunk: [dydz 9][cout half 2][dx 3]
constexpr int kUpFrags = 128;    // fragments of an upsampled chunk: [class (py,pz) 4][tytz 4][cout half 2][px 2][tx 2]

struct UpfArgs {
    const char* skip;            // (B, Xt, Yt, Zt, skipC) fp16, activated
    const char* up;              // (B, Xt/2, Yt/2, Zt/2, upC) fp16, activated
    long long skip_plane, skip_batch, up_plane, up_batch;   // bytes
    int skipC, upC;
    int ns, nu;                  // 32-channel chunks of the two sources
    const char* wpk;
    const float* bias;
    char* out;
    float* partial;
    int B, Xt, Yt, Zt, Yl, Zl;
    int XC, nxc, npatch;
    int K;                       // low-resolution rows per workgroup
    int SUBP, nposp, nposl;      // positions per fine sub-plane / fine plane / low-resolution plane
    int out_vs, cout_off;        // bytes per output voxel line; first output channel of this launch (a launch computes 32)
    int pstride, poff;           // floats per gn_partial row (cout/4 * 2); offset of this launch's 16 in it
    long long* dbg;              // -DSK_TIMING builds: per-wave phase cycle sums (tools/upfold_phase_timing.py)
    int w8_off, w8_scale, wpk_bytes;   // MIX8: byte offset of the fp8 fragments, E8M0 scale of the weights (x 4), bytes of the image
};

constexpr int kSkipFrags8 = 30;  // MIX8, 2 KiB fp8 fragments of a skip chunk: [tap-row pair 5][cout half 2][dx 3]
constexpr int kUpFrags8 = 64;    // of an upsampled chunk: [class 4][ty 2][cout half 2][px 2][tx 2] (K = two tz taps)

// Phase timing (-DSK_TIMING build only): per wave, cycles between the marks of a phase
#ifdef SK_TIMING
#define SK_T_DECL long long tacc_[sk::kTimingSlots] = {0}; long long tprev_ = __builtin_readcyclecounter();
#define SK_T(i) { const long long t_ = __builtin_readcyclecounter(); tacc_[i] += t_ - tprev_; tprev_ = t_; }
#define SK_T_DUMP(a, w, lane) if ((a).dbg && blockIdx.x < sk::kTimingBlocks && (w) < 4 && (lane) == 0) { \
        for (int i_ = 0; i_ < sk::kTimingSlots; ++i_) (a).dbg[((long long)blockIdx.x * 4 + (w)) * sk::kTimingSlots + i_] = tacc_[i_]; }
#else
#define SK_T_DECL
#define SK_T(i)
#define SK_T_DUMP(a, w, lane)
#endif

#define SK_UPF_NAME conv3_upf_kernel
#define SK_UPF_MIX8 0
#include "conv3d_up_kernel.inc"
#undef SK_UPF_NAME
#undef SK_UPF_MIX8
#define SK_UPF_NAME conv3_upf_mix8_kernel
#define SK_UPF_MIX8 1
#include "conv3d_up_kernel.inc"
#undef SK_UPF_NAME
#undef SK_UPF_MIX8

struct UpfPlan {
    int K, SUBP, nposp, nposl, npatch, XC, nxc;
    size_t lds;
};

// 0: supported.  The folded kernel covers the geometries whose low-resolution rows fit a 32-column segment.
int make_upf_plan(UpfPlan& p, int Xt, int Yt, int Zt) {
    if (Xt % 2 || Yt % 2 || Zt % 2) return -1;
    const int Zl = Zt / 2, Yl = Yt / 2;
    if (Zl < 1 || Zl > 32) return -1;
    p.K = 32 / Zl;
    const int over = 32 - p.K * Zl;                       // columns past the segment still form addresses
    p.SUBP = (p.K + 1) * Zl + 2;
    p.nposp = (4 * p.SUBP + over + 15) / 16 * 16;
    p.nposl = ((p.K + 2) * Zl + 2 + over + 15) / 16 * 16;
    if (p.nposp > 64 * kMaxDma || p.nposl > 64 * kMaxDmaL || p.nposl > p.nposp) return -1;
    p.lds = (size_t)6 * (p.nposp + sk::kZeroPos) * kPosBytes;
    if (p.lds > 80 * 1024) return -1;
    p.npatch = (Yl + p.K - 1) / p.K;
    // x-chunks as conv3d.hip's make_plan: a function of the tile geometry only (batch-invariant bits)
    const int kPlanBatch = 8, xs = 4;
    const int target = 256 * 2 * 6;
    int nxc = (target + p.npatch * kPlanBatch - 1) / (p.npatch * kPlanBatch);
    const int max_nxc = (Xt + 2 * xs - 1) / (2 * xs);
    if (nxc > max_nxc) nxc = max_nxc;
    if (nxc < 1) nxc = 1;
    int XC = (Xt + nxc - 1) / nxc;
    XC = (XC + xs - 1) / xs * xs;
    p.XC = XC;
    p.nxc = (Xt + XC - 1) / XC;
    return 0;
}

}  // namespace

extern "C" {

int sk_conv3d_upfold_num_blocks(int ox, int oy, int oz, int cout) {
    UpfPlan p;
    if ((cout != 32 && cout != 64) || make_upf_plan(p, ox, oy, oz)) return -1;
    return p.npatch * p.nxc;
}

static int64_t pack_upfold(const float* w, int cout, int c_skip, int c_up, void* dst, bool split) {
    if ((cout != 32 && cout != 64) || c_skip <= 0 || c_up <= 0 || c_skip % 32 || c_up % 32) {
        sk::set_error("sk_conv3d_pack_weight_upfold_host: unsupported shape cout=%d c_skip=%d c_up=%d", cout, c_skip, c_up);
        return SK_ERR_ARG;
    }
    const int ns = c_skip / 32, nu = c_up / 32, cin = c_skip + c_up;
    const int nsets = split ? 2 : 1;   // split: per chunk the lo-weight fragments, then the hi-weight fragments
    const int64_t nfrag = ((int64_t)ns * kSkipFrags + (int64_t)nu * kUpFrags) * nsets * (cout / 32);
    if (!dst) return nfrag * 1024;
    t16* out = (t16*)dst;
    auto W = [&](int co, int ci, int kx, int ky, int kz) { return w[((((int64_t)co * cin + ci) * 3 + kx) * 3 + ky) * 3 + kz]; };
    // plain: fp16(v); split set 0: lo = fp16(v - fp16(v)), set 1: hi = fp16(v)
    auto part = [&](double v, int set) {
        const t16 hi = (t16)(float)v;
        if (!split || set == 1) return hi;
        return (t16)(float)(v - (double)(float)hi);
    };
    int64_t f = 0;
    // the folded weight of parity p, tap t along an axis sums the kernel taps k (0..2) that read the same low-resolution
    // voxel: p=0: t=0 {0}, t=1 {1,2}; p=1: t=0 {0,1}, t=1 {2}
    auto lo = [](int p, int t) { return p == 0 ? (t == 0 ? 0 : 1) : (t == 0 ? 0 : 2); };
    auto hi = [](int p, int t) { return p == 0 ? (t == 0 ? 0 : 2) : (t == 0 ? 1 : 2); };
    for (int cg = 0; cg < cout / 32; ++cg) {   // one fragment set per 32 output channels (= per launch of the kernel)
        // skip chunks: conv3_m16_kernel's order [chunk][dy*3+dz][cout half][dx]; lane l holds W[16 i + (l&15)][c0 + 8 (l>>4) + e]
        for (int ch = 0; ch < ns; ++ch)
            for (int set = 0; set < nsets; ++set)
                for (int dydz = 0; dydz < 9; ++dydz)
                    for (int i = 0; i < 2; ++i)
                        for (int dx = 0; dx < 3; ++dx, ++f)
                            for (int l = 0; l < 64; ++l)
                                for (int e = 0; e < 8; ++e)
                                    out[f * 512 + l * 8 + e] =
                                        part(W(32 * cg + 16 * i + (l & 15), ch * 32 + 8 * (l >> 4) + e, dx, dydz / 3, dydz % 3), set);
        // upsampled chunks: [chunk][class 2 py + pz][ty*2+tz][cout half][px][tx]; the fold is summed in double, then split
        for (int ch = 0; ch < nu; ++ch)
            for (int set = 0; set < nsets; ++set)
                for (int cls = 0; cls < 4; ++cls)
                    for (int tytz = 0; tytz < 4; ++tytz)
                        for (int i = 0; i < 2; ++i)
                            for (int px = 0; px < 2; ++px)
                                for (int tx = 0; tx < 2; ++tx, ++f) {
                                    const int py = cls >> 1, pz = cls & 1, ty = tytz >> 1, tz = tytz & 1;
                                    for (int l = 0; l < 64; ++l)
                                        for (int e = 0; e < 8; ++e) {
                                            const int co = 32 * cg + 16 * i + (l & 15), ci = c_skip + ch * 32 + 8 * (l >> 4) + e;
                                            double sacc = 0.0;
                                            for (int kx = lo(px, tx); kx <= hi(px, tx); ++kx)
                                                for (int ky = lo(py, ty); ky <= hi(py, ty); ++ky)
                                                    for (int kz = lo(pz, tz); kz <= hi(pz, tz); ++kz) sacc += (double)W(co, ci, kx, ky, kz);
                                            out[f * 512 + l * 8 + e] = part(split ? sacc : (double)(float)sacc, set);
                                        }
                                }
    }
    return nfrag * 1024;
}

int64_t sk_conv3d_pack_weight_upfold_host(const float* w, int cout, int c_skip, int c_up, void* dst) {
    return pack_upfold(w, cout, c_skip, c_up, dst, false);
}

int64_t sk_conv3d_pack_weight_upfold_split_host(const float* w, int cout, int c_skip, int c_up, void* dst) {
    return pack_upfold(w, cout, c_skip, c_up, dst, true);
}

// precision "mix8": per 32 output channels [the fp16 fragments of the hi weights, as sk_conv3d_pack_weight_upfold_host lays them
// out][fp8 fragments of 2 KiB: skip chunks [chunk][tap-row pair 5][cout half 2][dx 3] -- lane l: cout 16 i + (l & 15), K block
// g = l >> 4: row 2 rp + (g >> 1), g & 1 = 0: e4m3(2^(b+11) w_lo) | 1: e4m3(2^b w) -- then upsampled chunks [chunk][class 4][ty 2]
// [cout half 2][px 2][tx 2] -- K block g: tap tz = g >> 1, g & 1 as above, of the FOLDED weight (summed in double; w_lo = sum -
// fp16(sum))].  b = *scale_exp: the largest power of two with 2^b max(|w|, |folded sums|) <= 240.
int64_t sk_conv3d_pack_weight_upfold_mix8_host(const float* w, int cout, int c_skip, int c_up, void* dst, int* scale_exp) {
    const int64_t f16b = pack_upfold(w, cout, c_skip, c_up, nullptr, false);
    if (f16b < 0) return f16b;
    const int ns = c_skip / 32, nu = c_up / 32, cin = c_skip + c_up, ncg = cout / 32;
    const int64_t f16_cg = f16b / ncg, f8_cg = ((int64_t)ns * kSkipFrags8 + (int64_t)nu * kUpFrags8) * 2048;
    const int64_t total = (f16_cg + f8_cg) * ncg;
    if (!dst) return total;
    auto W = [&](int co, int ci, int kx, int ky, int kz) { return w[((((int64_t)co * cin + ci) * 3 + kx) * 3 + ky) * 3 + kz]; };
    auto lo = [](int p, int t) { return p == 0 ? (t == 0 ? 0 : 1) : (t == 0 ? 0 : 2); };
    auto hi = [](int p, int t) { return p == 0 ? (t == 0 ? 0 : 2) : (t == 0 ? 1 : 2); };
    auto folded = [&](int co, int ci, int px, int tx, int py, int ty, int pz, int tz) {
        double sacc = 0.0;
        for (int kx = lo(px, tx); kx <= hi(px, tx); ++kx)
            for (int ky = lo(py, ty); ky <= hi(py, ty); ++ky)
                for (int kz = lo(pz, tz); kz <= hi(pz, tz); ++kz) sacc += (double)W(co, ci, kx, ky, kz);
        return sacc;
    };
    double wmax = 0.0;
    for (int co = 0; co < cout; ++co) {
        for (int ci = 0; ci < c_skip; ++ci)
            for (int t = 0; t < 27; ++t) wmax = std::fmax(wmax, std::fabs((double)W(co, ci, t / 9, (t / 3) % 3, t % 3)));
        for (int ci = c_skip; ci < cin; ++ci)
            for (int c = 0; c < 64; ++c)
                wmax = std::fmax(wmax, std::fabs(folded(co, ci, c & 1, (c >> 1) & 1, (c >> 2) & 1, (c >> 3) & 1, (c >> 4) & 1, (c >> 5) & 1)));
    }
    int b = 0;
    while (b < 40 && std::ldexp(wmax, b + 1) <= 240.0) ++b;
    if (scale_exp) *scale_exp = b;
    auto enc = [&](double v, int kind) {   // kind 0: the lo part at 2^(b+11), 1: the value at 2^b
        const double h = (double)(float)(t16)(float)v;
        return sk::f32_to_e4m3((float)std::ldexp(kind ? v : v - h, kind ? b : b + 11));
    };
    std::vector<char> f16(f16b);
    if (pack_upfold(w, cout, c_skip, c_up, f16.data(), false) != f16b) return SK_ERR_ARG;
    for (int cg = 0; cg < ncg; ++cg) {
        char* img = reinterpret_cast<char*>(dst) + cg * (f16_cg + f8_cg);
        memcpy(img, f16.data() + cg * f16_cg, (size_t)f16_cg);
        uint8_t* o = reinterpret_cast<uint8_t*>(img + f16_cg);
        int64_t f = 0;
        for (int ch = 0; ch < ns; ++ch)
            for (int rp = 0; rp < 5; ++rp)
                for (int i = 0; i < 2; ++i)
                    for (int dx = 0; dx < 3; ++dx, ++f)
                        for (int l = 0; l < 64; ++l)
                            for (int j = 0; j < 32; ++j) {
                                const int g = l >> 4, row = 2 * rp + (g >> 1);
                                o[f * 2048 + l * 32 + j] =
                                    row < 9 ? enc((double)W(32 * cg + 16 * i + (l & 15), ch * 32 + j, dx, row / 3, row % 3), g & 1) : 0;
                            }
        for (int ch = 0; ch < nu; ++ch)
            for (int cls = 0; cls < 4; ++cls)
                for (int ty = 0; ty < 2; ++ty)
                    for (int i = 0; i < 2; ++i)
                        for (int px = 0; px < 2; ++px)
                            for (int tx = 0; tx < 2; ++tx, ++f)
                                for (int l = 0; l < 64; ++l)
                                    for (int j = 0; j < 32; ++j) {
                                        const int g = l >> 4;
                                        const double v = folded(32 * cg + 16 * i + (l & 15), c_skip + ch * 32 + j, px, tx, cls >> 1, ty, cls & 1, g >> 1);
                                        o[f * 2048 + l * 32 + j] = enc(v, g & 1);
                                    }
    }
    return total;
}

static int upfold_impl(const void* skip, int c_skip, const void* up, int c_up, const void* weight, const float* bias,
                       void* out, int B, int ox, int oy, int oz, int cout, float* gn_partial, void* stream_, const bool split,
                       const bool mix8 = false, const int w8_scale_exp = 0) {
    hipStream_t stream = (hipStream_t)stream_;
    SK_CHECK_ARG(skip && up && weight && bias && out, "sk_conv3d_upfold: NULL pointer");
    SK_CHECK_ARG(cout == 32 || cout == 64, "sk_conv3d_upfold: cout must be 32 or 64 (got %d)", cout);
    SK_CHECK_ARG(c_skip > 0 && c_up > 0 && c_skip % 32 == 0 && c_up % 32 == 0 && c_skip + c_up <= 256,
                 "sk_conv3d_upfold: channel counts must be multiples of 32 (got %d + %d)", c_skip, c_up);
    SK_CHECK_ARG(B > 0 && ox > 0 && oy > 0 && oz > 0, "sk_conv3d_upfold: bad output extents");
    UpfPlan p;
    SK_CHECK_ARG(make_upf_plan(p, ox, oy, oz) == 0,
                 "sk_conv3d_upfold: geometry (%d,%d,%d) unsupported (sk_conv3d_upfold_num_blocks < 0: use sk_conv3d)", ox, oy, oz);
    UpfArgs a{};
    a.skip = (const char*)skip;
    a.up = (const char*)up;
    a.skipC = c_skip;
    a.upC = c_up;
    const int lanes = split ? 2 : 1;   // fp16 values per logical channel in a voxel line: [hi | lo]
    a.skip_plane = (long long)oy * oz * c_skip * 2 * lanes;
    a.skip_batch = a.skip_plane * ox;
    a.up_plane = (long long)(oy / 2) * (oz / 2) * c_up * 2 * lanes;
    a.up_batch = a.up_plane * (ox / 2);
    a.ns = c_skip / 32;
    a.nu = c_up / 32;
    a.wpk = (const char*)weight;
    a.bias = bias;
    a.out = (char*)out;
    a.partial = gn_partial;
    a.B = B;
    a.Xt = ox;
    a.Yt = oy;
    a.Zt = oz;
    a.Yl = oy / 2;
    a.Zl = oz / 2;
    a.XC = p.XC;
    a.nxc = p.nxc;
    a.npatch = p.npatch;
    a.K = p.K;
    a.SUBP = p.SUBP;
    a.nposp = p.nposp;
    a.nposl = p.nposl;
    a.dbg = nullptr;
#ifdef SK_TIMING
    a.dbg = sk::timing_buffer();
#endif
    const bool wlds = !split && a.ns == 1 && p.lds + 12288 <= 80 * 1024;   // two tap rows of the single skip chunk in LDS
    auto kern = mix8 ? conv3_upf_mix8_kernel<4, false, true>
                     : (split ? conv3_upf_kernel<4, false, true> : (wlds ? conv3_upf_kernel<4, true, false> : conv3_upf_kernel<4, false, false>));
    const size_t f16_cg = (size_t)(a.ns * kSkipFrags + a.nu * kUpFrags) * 1024, f8_cg = (size_t)(a.ns * kSkipFrags8 + a.nu * kUpFrags8) * 2048;
    if (mix8) {
        SK_CHECK_ARG(split && w8_scale_exp >= 0 && w8_scale_exp < 64, "sk_conv3d_upfold_mix8: bad weight scale exponent %d", w8_scale_exp);
        // a low-resolution plane holds both halves of its lines in one ring slot, in front of the slot's zero window
        SK_CHECK_ARG(2 * p.nposl <= p.nposp, "sk_conv3d_upfold_mix8: geometry (%d,%d,%d) unsupported", ox, oy, oz);
        a.w8_off = (int)f16_cg;
        a.wpk_bytes = (int)(f16_cg + f8_cg);
        a.w8_scale = 0x01010101 * (127 - w8_scale_exp);
    }
    const size_t lds = p.lds + (wlds ? 12288 : 0);
    if (lds > 48 * 1024)
        SK_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)(p.npatch * p.nxc * B);
    // 32 output channels per launch (COUT 64: two launches over the same inputs -- twice the staging, but on the
    // 16x16x32 matrix instruction and at 70 instead of 108 tap-chunks; measured against conv3_kernel<64> in DESIGN.md)
    a.out_vs = cout * 2 * lanes;
    a.pstride = (cout / 4) * 2;
    for (int cg = 0; cg < cout / 32; ++cg) {
        a.cout_off = 32 * cg;
        a.poff = 16 * cg;
        a.wpk = (const char*)weight + (mix8 ? (size_t)cg * (f16_cg + f8_cg) : (size_t)cg * (a.ns * kSkipFrags + a.nu * kUpFrags) * lanes * 1024);
        kern<<<grid, 256, lds, stream>>>(a);
        SK_CHECK_LAUNCH();
    }
    return SK_OK;
}


int sk_conv3d_upfold(const void* skip, int c_skip, const void* up, int c_up, const void* weight, const float* bias,
                     void* out, int B, int ox, int oy, int oz, int cout, float* gn_partial, void* stream) {
    return upfold_impl(skip, c_skip, up, c_up, weight, bias, out, B, ox, oy, oz, cout, gn_partial, stream, false);
}

int sk_conv3d_upfold_split(const void* skip, int c_skip, const void* up, int c_up, const void* weight, const float* bias,
                           void* out, int B, int ox, int oy, int oz, int cout, float* gn_partial, void* stream) {
    return upfold_impl(skip, c_skip, up, c_up, weight, bias, out, B, ox, oy, oz, cout, gn_partial, stream, true);
}

int sk_conv3d_upfold_mix8(const void* skip, int c_skip, const void* up, int c_up, const void* weight, int weight_scale_exp,
                          const float* bias, void* out, int B, int ox, int oy, int oz, int cout, float* gn_partial, void* stream) {
    return upfold_impl(skip, c_skip, up, c_up, weight, bias, out, B, ox, oy, oz, cout, gn_partial, stream, true, true, weight_scale_exp);
}

}  // extern "C"
