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
};

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

// WLDS (one skip chunk only): tap rows 0 and 1 of the skip chunk's weights sit in LDS behind the ring, one copy per
// workgroup, instead of being streamed by each of the four waves (conv3_m16_kernel's WL): 12 of the 54 + 32 weight loads
// per wave and step become LDS reads and the hand-off to the skip phase requests no weights at all.
// SPLIT (precision "split", sk_conv3d_upfold_split): every tensor holds fp16 hi + lo pairs, [hi (C) | lo (C)] per voxel
// line, and every weight is a hi + lo pair; a logical chunk runs as three phases -- (x_hi, w_lo), (x_hi again: no LDS-DMA,
// w_hi), (x_lo, w_hi) -- as in conv3_m16_kernel's split mode, and the fp32 accumulator is stored as hi = fp16(v),
// lo = fp16(v - hi).  The folded weights are split AFTER the fold (the sum is formed in double on the host).
template <int XS, bool WLDS, bool SPLIT>
__global__ void __launch_bounds__(256, 2) conv3_upf_kernel(UpfArgs a) {
    constexpr int R = XS + 2;          // fine planes of a step
    constexpr int RL = XS / 2 + 2;     // low-resolution planes of a step
    static_assert(XS == 4, "x parity of an output plane must be a compile-time constant");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int py = w >> 1, pz = w & 1;   // parity class of this wave's output voxels
    const int c16 = lane & 15, g = lane >> 4;

    // XCD-aware workgroup order (conv3d.hip)
    int blk = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = blk & 7, qn = nwg >> 3, rn = nwg & 7;
        blk = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blk >> 3);
    }
    const int patch = blk % a.npatch;
    blk /= a.npatch;
    const int xc = blk % a.nxc;
    const int b = blk / a.nxc;
    const int block_in_batch = xc * a.npatch + patch;
    const int nblk = a.npatch * a.nxc;

    const int Zl = a.Zl, Zt = a.Zt;
    const int yl0 = patch * a.K;             // first low-resolution row of this workgroup
    const int nseg = a.K * Zl;               // columns in use (<= 32)

    // per-lane flags of column (j): bit j: the z-1 tap wraps (class pz = 0, zl = 0) | bit 8 + j: the z+1 tap wraps
    // (pz = 1, zl = Zl-1) | bit 16 + j: a real voxel
    unsigned vflags = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = 16 * j + c16;
        const int r = m / Zl, c = m - r * Zl;
        vflags |= (unsigned)(pz == 0 && c == 0) << j;
        vflags |= (unsigned)(pz == 1 && c == Zl - 1) << (8 + j);
        vflags |= (unsigned)(m < nseg && yl0 + r < a.Yl) << (16 + j);
    }
    auto zlo = [&](int j) { return (vflags >> j) & 1u; };
    auto zhi = [&](int j) { return (vflags >> (8 + j)) & 1u; };
    auto vvalid = [&](int j) { return (vflags >> (16 + j)) & 1u; };

    // the voxel this lane STORES (after the permlane transpose it owns 8 channels of column c16 + 16 (g & 1))
    int ovox;      // in-plane fine voxel index, -1: none
    {
        const int m = c16 + 16 * (g & 1);
        const int r = m / Zl, c = m - r * Zl;
        ovox = (m < nseg && yl0 + r < a.Yl) ? (2 * (yl0 + r) + py) * Zt + 2 * c + pz : -1;
    }

    // ---- LDS-DMA bookkeeping ------------------------------------------------------------
    // fine plane: position q = sub * SUBP + 1 + ylr * Zl + zl, sub = 2 (y & 1) + (z & 1), rows ylr = (y >> 1) - (yl0 - (y & 1))
    const int ndma = a.nposp / 16, ndmal = a.nposl / 16;
    const int d_cs = ((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16;   // source chunk of this lane's slot (conv3_m16_kernel)
    int d_vox[kMaxDma], d_low[kMaxDmaL];
#pragma unroll
    for (int k = 0; k < kMaxDma; ++k) {
        const int t = w + 4 * k;
        const int q = (64 * t + lane) >> 2;
        const int sub = q / a.SUBP, rem = q - sub * a.SUBP - 1;
        const int ylr = rem >= 0 ? rem / Zl : 0, zl = rem - ylr * Zl;
        const int yp = sub >> 1, zp = sub & 1;
        const int y = 2 * (yl0 - yp + ylr) + yp, z = 2 * zl + zp;
        const bool ok = t < ndma && sub < 4 && rem >= 0 && ylr <= a.K && y >= 0 && y < a.Yt;
        d_vox[k] = ok ? y * Zt + z : -1;
    }
    // low-resolution plane: position q = 1 + (yl - (yl0 - 1)) * Zl + zl
#pragma unroll
    for (int k = 0; k < kMaxDmaL; ++k) {
        const int t = w + 4 * k;
        const int q = (64 * t + lane) >> 2;
        const int rem = q - 1;
        const int ylr = rem >= 0 ? rem / Zl : 0, zl = rem - ylr * Zl;
        const int yl = yl0 - 1 + ylr;
        const bool ok = t < ndmal && rem >= 0 && ylr < a.K + 2 && yl >= 0 && yl < a.Yl;
        d_low[k] = ok ? yl * Zl + zl : -1;
    }

    f32x4 acc[XS][2][2];   // [output plane][cout half i][voxel half j]
    float gsum[2], gsq[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) gsum[q] = gsq[q] = 0.0f;

    const int xa = xc * a.XC;
    const int xb = min(xa + a.XC, a.Xt);
    const int Xl = a.Xt >> 1;
    const int plane_bytes = (a.nposp + sk::kZeroPos) * kPosBytes;
    const int zero_addr = a.nposp * kPosBytes;   // never written by either image's DMA (nposl <= nposp)
    const long long out_plane = (long long)a.Yt * Zt * a.out_vs;
    char* outb = a.out + (long long)b * a.Xt * out_plane;
    const float* biasp = a.bias + a.cout_off + 4 * g;

    const int nsteps = (xb - xa + XS - 1) / XS;
    // Virtual chunks: plain mode one per 32-channel chunk; split mode three -- part 0: hi halves x lo weights, part 1: the
    // SAME staged planes x hi weights (no LDS-DMA), part 2: lo halves x hi weights
    constexpr int kParts = SPLIT ? 3 : 1;
    const int nvs = a.ns * kParts, nvu = a.nu * kParts;
    auto v_chunk = [](int v) { return SPLIT ? v / 3 : v; };
    auto v_part = [](int v) { return SPLIT ? v % 3 : 0; };
    auto v_dma = [&](int v) { return v_part(v) != 1; };

    // Phase order of a step: the skip chunks, then the upsampled chunks.  Fine plane i of the step lives in ring slot
    // (rot + i) % R; the low-resolution planes of the step take the slots of fine planes 0 .. RL-1, which the step is
    // done with -- fine planes XS, XS + 1 stay staged through the upsampled phases and are planes 0, 1 of the next
    // step's skip chunk when there is only one (`reuse`: XS instead of XS + 2 plane loads per step).
    auto issue_fine = [&](int step, int chs, bool reuse, int rot_n) {
        const int x0 = xa + step * XS;
        const int xlo = max(x0 - 1, 0);
        const long long wbytes = min((long long)(R + 1) * a.skip_plane, a.skip_batch - (long long)xlo * a.skip_plane);
        const __amdgpu_buffer_rsrc_t rsrc = sk::make_rsrc(a.skip + (long long)b * a.skip_batch + (long long)xlo * a.skip_plane, (unsigned)wbytes);
        const unsigned vstride = (unsigned)(a.skipC * 2 * (SPLIT ? 2 : 1));
        const int choff = v_chunk(chs) * 64 + (v_part(chs) == 2 ? a.skipC * 2 : 0);   // split: the lo halves follow the hi halves
        for (int i = reuse ? 2 : 0; i < R; ++i) {
            const int x = x0 - 1 + i;
            const bool xok = x >= 0 && x < a.Xt;
            char* lbase = lds + ((rot_n + i) % R) * plane_bytes;
            const unsigned xoff = (unsigned)((x - xlo) * (int)a.skip_plane + choff + d_cs);
#pragma unroll
            for (int k = 0; k < kMaxDma; ++k) {
                const int t = w + 4 * k;
                if (t < ndma) {
                    int dv = d_vox[k];
                    asm volatile("" : "+v"(dv));   // keep the offset arithmetic here: hoisted out of the step loop it costs registers
                    const unsigned voff = (xok && dv >= 0) ? xoff + (unsigned)dv * vstride : sk::kOob;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lbase + t * 1024), 16, voff, 0, 0, 0);
                }
            }
        }
    };
    auto issue_low = [&](int step, int chu, bool /*reuse: never -- the skip phases overwrite these slots*/, int rotl_n) {
        const int xl0 = ((xa + step * XS) >> 1) - 1;
        const int xlo = max(xl0, 0);
        const long long wbytes = min((long long)(RL + 1) * a.up_plane, a.up_batch - (long long)xlo * a.up_plane);
        const __amdgpu_buffer_rsrc_t rsrc = sk::make_rsrc(a.up + (long long)b * a.up_batch + (long long)xlo * a.up_plane, (unsigned)wbytes);
        const unsigned vstride = (unsigned)(a.upC * 2 * (SPLIT ? 2 : 1));
        const int choff = v_chunk(chu) * 64 + (v_part(chu) == 2 ? a.upC * 2 : 0);
        for (int i = 0; i < RL; ++i) {
            const int xl = xl0 + i;
            const bool xok = xl >= 0 && xl < Xl;
            char* lbase = lds + ((rotl_n + i) % R) * plane_bytes;
            const unsigned xoff = (unsigned)((xl - xlo) * (int)a.up_plane + choff + d_cs);
#pragma unroll
            for (int k = 0; k < kMaxDmaL; ++k) {
                const int t = w + 4 * k;
                if (t < ndmal) {
                    const unsigned voff = (xok && d_low[k] >= 0) ? xoff + (unsigned)d_low[k] * vstride : sk::kOob;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lbase + t * 1024), 16, voff, 0, 0, 0);
                }
            }
        }
    };
    const __amdgpu_buffer_rsrc_t wrsrc = sk::make_rsrc(a.wpk, (unsigned)((a.ns * kSkipFrags + a.nu * kUpFrags) * (SPLIT ? 2 : 1) * 1024));
    const unsigned wlane = lane * 16;
    auto wload = [&](unsigned off) {
        return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, __builtin_amdgcn_readfirstlane(off), 0));
    };
    // first fragment of a chunk for this wave
    // split: per chunk the lo-weight fragment set, then the hi-weight set; part 0 multiplies by lo, parts 1 and 2 by hi
    auto wbase_skip = [&](int v) {
        return (unsigned)((SPLIT ? (2 * v_chunk(v) + (v_part(v) != 0)) : v) * kSkipFrags * 1024);
    };
    auto wbase_up = [&](int v) {
        const int frag0 = SPLIT ? 2 * a.ns * kSkipFrags + (2 * v_chunk(v) + (v_part(v) != 0)) * kUpFrags
                                : a.ns * kSkipFrags + v * kUpFrags;
        return (unsigned)((frag0 + w * 32) * 1024);
    };

    // Weight fragments that cross a phase boundary (requested before the closing barrier of the phase before): ONE
    // register set for both kinds of chunk -- a value carried around the phase loop stays allocated through every phase.
    // skip chunk: wq[ks * 3 + dx] = the two cout halves of a tap row; upsampled chunk: wq[(i*2 + px)*2 + tx]
    // of a (ty, tz) row.
    half8 wq[8];
    auto load_af = [&](unsigned wch, int tytz, half8 (&dst)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = wload(wch + (unsigned)((tytz * 8 + e) * 1024));
    };
    const char* wlds = lds + R * plane_bytes + lane * 16;   // WLDS: fragments 0 .. 11 of the skip chunk
    auto prefetch_skip = [&](int cs) {   // the first fragments of a phase
        if constexpr (WLDS) return;      // row 0 comes from LDS at the start of the phase
#pragma unroll
        for (int e = 0; e < 6; ++e) wq[e] = wload(wbase_skip(cs) + e * 1024);
    };
    half8 wq1[8];   // upsampled chunk: second tap row (live from the hand-off before the phase to its second row only)
    auto prefetch_up = [&](int cu) {
        load_af(wbase_up(cu), 0, wq);
        load_af(wbase_up(cu), 1, wq1);
    };

    if (tid < R * 4 * sk::kZeroPos)
        *reinterpret_cast<uint4*>(lds + (tid / (4 * sk::kZeroPos)) * plane_bytes + zero_addr + (tid % (4 * sk::kZeroPos)) * 16) = make_uint4(0, 0, 0, 0);
    if constexpr (WLDS) {
        for (int i = tid; i < 12 * 64; i += 256)
            *reinterpret_cast<uint4*>(lds + R * plane_bytes + i * 16) = *reinterpret_cast<const uint4*>(a.wpk + i * 16);
    }
    issue_fine(0, 0, false, 0);
    prefetch_skip(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const bool ring = !SPLIT && a.ns == 1;   // the same skip chunk every step: its two trailing planes are reused
    int rot = 0;
    SK_T_DECL
    for (int step = 0; step < nsteps; ++step) {
        const int x0 = xa + step * XS;
#pragma unroll
        for (int o = 0; o < XS; ++o)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[o][i][0] = acc[o][i][1] = *reinterpret_cast<const f32x4*>(biasp + 16 * i);
        // The LDS addresses of the taps are cheap functions of the lane's column: recomputed every step.  Without this
        // opaque copy the compiler hoists all 18 + 8 of them out of the step loop and spills.
        int c16v = c16;
        asm volatile("" : "+v"(c16v));

        for (int cs = 0; cs < nvs; ++cs) {
            const unsigned wch = wbase_skip(cs);
            // ---------------- skip chunk: 27 taps on the de-interleaved fine planes -----------------
            int pslot[R];
#pragma unroll
            for (int i = 0; i < R; ++i) pslot[i] = ((rot + i) % R) * plane_bytes;
            // A body = (tap row, voxel half j): R fragments feeding 6 XS MFMAs (both cout halves); the next body's fragments
            // and the NEXT ROW's six weight fragments are requested before this body's MFMAs.
            half8 bb[2][R];
            half8 wb[6];
            auto load_body = [&](int dydz, int j, half8 (&dst)[R]) {
                const int dy = dydz / 3 - 1, dz = dydz % 3 - 1;
                // fine (y, z) = (2 yl + py + dy, 2 zl + pz + dz): sub-plane by the parities, position offset by the halves
                const int Y = py + dy, Z = pz + dz;
                const int tapoff = ((Y & 1) * 2 + (Z & 1)) * a.SUBP + 1 + (Y >= 1 ? Zl : 0) + (Z >> 1);   // Z >> 1: floor
                const int q = c16v + tapoff;
                int addr = (q * 4 + (g ^ (((q >> 2) & 1) << 1))) * 16 + 1024 * j;   // (q + 16) >> 2 has the parity of q >> 2
                if (dz < 0) addr = zlo(j) ? sk::zero_of(zero_addr, addr) : addr;
                if (dz > 0) addr = zhi(j) ? sk::zero_of(zero_addr, addr) : addr;
#pragma unroll
                for (int i = 0; i < R; ++i) dst[i] = *reinterpret_cast<const half8*>(lds + pslot[i] + addr);
            };
            auto mma_body = [&](int j, const half8* wf, const half8 (&src)[R]) {
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            const int o = i - d;
                            if (o >= 0 && o < XS) acc[o][ks][j] = SK_MFMA_16x16x32_T16(wf[ks * 3 + d], src[i], acc[o][ks][j], 0, 0, 0);
                        }
            };
            if constexpr (WLDS) {
#pragma unroll
                for (int e = 0; e < 6; ++e) wq[e] = *reinterpret_cast<const half8*>(wlds + e * 1024);
            }
            load_body(0, 0, bb[0]);
#pragma unroll
            for (int dydz = 0; dydz < 9; ++dydz) {
                half8* wcur = (dydz & 1) ? wb : wq;
                half8* wnext = (dydz & 1) ? wq : wb;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 0) {
                        load_body(dydz, 1, bb[1]);
                        if (dydz < 8) {
#pragma unroll
                            for (int e = 0; e < 6; ++e) {
                                if (WLDS && dydz + 1 < 2)
                                    wnext[e] = *reinterpret_cast<const half8*>(wlds + ((dydz + 1) * 6 + e) * 1024);
                                else
                                    wnext[e] = wload(wch + (unsigned)(((dydz + 1) * 6 + e) * 1024));
                            }
                        }
                    } else if (dydz < 8) {
                        load_body(dydz + 1, 0, bb[0]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mma_body(j, wcur, bb[j]);
                }
            }
            SK_T(0)   // skip MFMA loop
            if (cs + 1 < nvs) {
                // hand the planes to the next skip chunk: barrier (all waves done reading) -> LDS-DMA + first weights of the
                // next phase -> landed -> barrier; split part 1 multiplies the planes that are staged: weights only
                if (v_dma(cs + 1)) {
                    __syncthreads();
                    issue_fine(step, cs + 1, false, rot);
                    prefetch_skip(cs + 1);
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                } else {
                    prefetch_skip(cs + 1);
                }
            }
        }
        // ... and to the first upsampled chunk (outside the loop: the weight rows requested here must not look live
        // through the skip chunks' MFMA loops)
        __syncthreads();
        SK_T(1)   // barrier
        issue_low(step, 0, false, rot);
        prefetch_up(0);
        SK_T(2)   // LDS-DMA issue + weight prefetch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SK_T(3)   // landing wait
        asm volatile("s_barrier" ::: "memory");
        SK_T(4)   // barrier

        for (int cu = 0; cu < nvu; ++cu) {
            const unsigned wch = wbase_up(cu);
            // ---------------- upsampled chunk: 2 x 2 x 2 folded taps on the low-resolution planes -----------------
            int pslot[RL];
#pragma unroll
            for (int i = 0; i < RL; ++i) pslot[i] = ((rot + i) % R) * plane_bytes;
            // a body = (tap row (ty, tz), voxel half j): RL fragments feeding 4 XS MFMAs
            half8 bl[2][RL];
            auto load_low = [&](int tytz, int j, half8 (&dst)[RL]) {
                const int sy = (tytz >> 1) + py - 1, sz = (tytz & 1) + pz - 1;   // low-resolution offsets of this tap
                const int q = c16v + 1 + (1 + sy) * Zl + sz;
                int addr = (q * 4 + (g ^ (((q >> 2) & 1) << 1))) * 16 + 1024 * j;
                addr = (sz < 0 && zlo(j)) ? sk::zero_of(zero_addr, addr) : addr;
                addr = (sz > 0 && zhi(j)) ? sk::zero_of(zero_addr, addr) : addr;
#pragma unroll
                for (int i = 0; i < RL; ++i) dst[i] = *reinterpret_cast<const half8*>(lds + pslot[i] + addr);
            };
            // Weight fragments two tap rows ahead: a fragment here feeds 4 MFMAs (2 planes of its x parity x 2 voxel halves)
            // and the four waves stream four different sets, which mostly come from L2 -- one row (32 MFMAs) of lead left
            // the wave waiting on every row (the phase took 5.6x its MFMA cycles).  Rows 0, 1 were requested before the
            // phase's opening barrier (wq, wq1), row 2 goes out now, row 3 takes row 0's registers.
            half8 wq2[8];
            load_af(wch, 2, wq2);
            load_low(0, 0, bl[0]);
#pragma unroll
            for (int tytz = 0; tytz < 4; ++tytz) {
                const half8* af = tytz == 1 ? wq1 : (tytz == 2 ? wq2 : wq);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 0) {
                        load_low(tytz, 1, bl[1]);
                    } else if (tytz < 3) {
                        load_low(tytz + 1, 0, bl[0]);
                    }
                    if (tytz == 1 && j == 0) load_af(wch, 3, wq);   // row 0 is done with wq
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int o = 0; o < XS; ++o)
#pragma unroll
                        for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const int px = o & 1, il = (o >> 1) + tx + px;   // low plane of tap tx for output plane o
                                acc[o][i][j] = SK_MFMA_16x16x32_T16(af[(i * 2 + px) * 2 + tx], bl[j][il], acc[o][i][j], 0, 0, 0);
                            }
                }
            }
            const bool last = cu + 1 == nvu;
            const bool have_next = !last || step + 1 < nsteps;
            const int rot_n = (last && ring) ? (rot + XS) % R : rot;
            SK_T(5)   // upsampled MFMA loop
            __syncthreads();
            SK_T(6)   // barrier
            if (have_next) {
                if (!last) {
                    if (v_dma(cu + 1)) issue_low(step, cu + 1, false, rot);
                    prefetch_up(cu + 1);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    issue_fine(step + 1, 0, ring, rot_n);
                    prefetch_skip(0);
                }
            }
            SK_T(7)   // LDS-DMA issue + weight prefetch
            if (last) {
                // ---------------- epilogue (overlaps the DMA): raw fp16 store + GroupNorm partials ------
                char* outw = outb + (long long)x0 * out_plane;
                const __amdgpu_buffer_rsrc_t rout = sk::make_rsrc(outw, (unsigned)(XS * out_plane));
    #pragma unroll
                for (int o = 0; o < XS; ++o) {
                    const int x = x0 + o;
    #pragma unroll
                    for (int part = 0; part < (SPLIT ? 2 : 1); ++part) {   // SPLIT: the hi halves, then the lo halves
    #pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        unsigned d[2][2];
    #pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const f32x4 r = acc[o][i][j];
                            half4 hv = {(t16)r[0], (t16)r[1], (t16)r[2], (t16)r[3]};
                            if (part == 1)   // lo = fp16(v - hi): exact difference, rounded once
                                hv = half4{(t16)(r[0] - (float)hv[0]), (t16)(r[1] - (float)hv[1]),
                                           (t16)(r[2] - (float)hv[2]), (t16)(r[3] - (float)hv[3])};
                            const uint2 u = __builtin_bit_cast(uint2, hv);
                            d[j][0] = u.x;
                            d[j][1] = u.y;
                            const bool in = vvalid(j) && x < xb;
                            if (SPLIT) {
                                if (part == 0 && in) {   // split: statistics of the fp32 accumulators
                                    gsum[i] += (r[0] + r[1]) + (r[2] + r[3]);
                                    gsq[i] += (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
                                }
                            } else {
                                const t16x2 z2 = {(t16)0.0f, (t16)0.0f}, one2 = {(t16)1.0f, (t16)1.0f};
                                const t16x2 lo2 = in ? t16x2{hv[0], hv[1]} : z2, hi2 = in ? t16x2{hv[2], hv[3]} : z2;
                                gsum[i] = SK_DOT2_T16(lo2, one2, gsum[i]);
                                gsum[i] = SK_DOT2_T16(hi2, one2, gsum[i]);
                                gsq[i] = SK_DOT2_T16(lo2, lo2, gsq[i]);
                                gsq[i] = SK_DOT2_T16(hi2, hi2, gsq[i]);
                            }
                        }
                        const auto s0 = __builtin_amdgcn_permlane16_swap(d[0][0], d[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(d[0][1], d[1][1], false, false);
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 lv = {s0[0], s1[0], s0[1], s1[1]};
                        const bool sok = x < xb && ovox >= 0;
                        // split: the voxel line is [hi (cout) | lo (cout)]: out_vs = 4 cout bytes, the lo halves cout * 2 bytes on
                        const unsigned off = (unsigned)(o * (int)out_plane + ovox * a.out_vs + part * (a.out_vs / 2) + a.cout_off * 2 +
                                                        32 * i + 16 * (g >> 1));
                        // always issued (the counted wait below relies on it); a masked lane's offset is out of range: dropped
                        __builtin_amdgcn_raw_buffer_store_b128(lv, rout, sok ? off : sk::kOob, 0, 0);
                    }
                    }
                }
                SK_T(8)   // epilogue
                if (have_next) {
                    // vmcnt retires in order: everything older than the epilogue's XS * 2 stores -- the LDS-DMA and the
                    // weight fragments -- has landed (conv3_m16_kernel)
                    constexpr int kStores = XS * 2 * (SPLIT ? 2 : 1);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kStores) : "memory");
                }
            }
            SK_T(9)   // deferred landing wait
            if (have_next) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            SK_T(10)  // closing barrier
            rot = rot_n;
        }
    }

    SK_T_DUMP(a, w, lane)
    // ---- block-level reduction of the GroupNorm partials ------------------------------------
    if (a.partial) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);  // [4 waves][8 quads][2]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float s = gsum[i], ss = gsq[i];
#pragma unroll
            for (int m = 8; m > 0; m >>= 1) {
                s += __shfl_xor(s, m);
                ss += __shfl_xor(ss, m);
            }
            if (c16 == 0) {
                red[(w * 8 + 4 * i + g) * 2 + 0] = s;
                red[(w * 8 + 4 * i + g) * 2 + 1] = ss;
            }
        }
        __syncthreads();
        if (tid < 16) {
            float t = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) t += red[q * 16 + tid];
            a.partial[((long long)b * nblk + block_in_batch) * a.pstride + a.poff + tid] = t;
        }
    }
}

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

static int upfold_impl(const void* skip, int c_skip, const void* up, int c_up, const void* weight, const float* bias,
                       void* out, int B, int ox, int oy, int oz, int cout, float* gn_partial, void* stream_, const bool split) {
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
    auto kern = split ? conv3_upf_kernel<4, false, true> : (wlds ? conv3_upf_kernel<4, true, false> : conv3_upf_kernel<4, false, false>);
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
        a.wpk = (const char*)weight + (size_t)cg * (a.ns * kSkipFrags + a.nu * kUpFrags) * lanes * 1024;
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

}  // extern "C"
