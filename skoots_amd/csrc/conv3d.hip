// 3x3x3 convolution (stride 1, zero pad 1) of the U-Net body as an implicit GEMM on
// the gfx950 matrix cores -- the dominant kernel of stage 1.
//
// Replaces the conv layers of the network the reference builds with
// cfg_to_bism_model (skoots/lib/utils.py:17-107) and runs under torch.compile +
// fp16 autocast at skoots/lib/eval.py:122-124,142-143.  Graph: oracle/unet_spec.py.
//
// Design (MI355X-first, not a translation of a cuDNN-style im2col):
//   * activations are channels-last fp16 (B, X, Y, Z, C), already normalised +
//     activated by the fused GroupNorm+SiLU pass; weights are pre-packed on the host
//     in MFMA A-fragment order (sk_conv3d_pack_weight_host);
//   * D[cout][voxel] += W[cout][cin,tap] * act[cin,tap][voxel]: rows = 32 output
//     channels, columns = 32 voxels of a (y,z) patch; conv3_kernel (COUT 64, 128) on
//     v_mfma_f32_32x32x16_f16 (K = 16 input channels of one tap), conv3_m16_kernel (COUT 32) on
//     v_mfma_f32_16x16x32_f16 (K = the tap's whole 32-channel chunk, 2 x 2 results per tile);
//   * a workgroup (4 waves, one 32-voxel column set each) owns a 128-voxel (y,z)
//     patch and MARCHES along x, XS output planes per step.  The input planes of a
//     step (XS+2, with y/z halo) sit in an LDS ring filled by LDS-DMA
//     (global_load_lds_dwordx4: no VGPRs, no VALU; halo / out-of-tile lanes read a
//     zero page, the nearest-neighbour upsample of the decoder is just the per-lane
//     source address); two planes are reused by the next step;
//   * one B fragment (ds_read_b128) feeds the three x-taps of three different output
//     planes, so LDS traffic is (XS+2)/(3*XS) reads per MFMA; the LDS image is
//     XOR-swizzled on the DMA *source* address (the destination is lane-linear) so the
//     16-lane groups of ds_read_b128 hit 16 distinct bank slots;
//   * input channels are consumed in chunks of 32 (the channel concat of skip +
//     upsampled tensors is two chunks from two sources, never materialised);
//   * the epilogue adds the bias, stores fp16 raw outputs (transposed through a per-wave LDS
//     pad into whole 64-byte voxel lines, 16 B per lane) and
//     accumulates per-channel-quad (sum, sumsq) of the fp32 accumulators for the
//     GroupNorm that follows; per-block partials are reduced in a fixed order by
//     sk_groupnorm_finalize (deterministic, no float atomics).
#include <stdlib.h>

#include <cmath>

#include <type_traits>
#include <vector>

#include "common.h"

namespace {

typedef t16 half8 __attribute__((ext_vector_type(8)));
typedef t16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kChunk = 32;       // input channels per LDS image
constexpr int kPosBytes = 64;    // kChunk * sizeof(fp16)
constexpr int kPatch = 128;      // voxels per (y,z) patch = 4 waves x 32 columns
constexpr int kMaxDma = 4;       // DMA wave-instructions per plane per wave (nposp <= 256)
constexpr int kPadStride = 64;   // bytes per voxel in the epilogue transpose pad (16-B chunks XOR-swizzled)
constexpr int kPadBytes = 32 * kPadStride;
constexpr int kMaxChunks = 24;   // phase chunks per step: 8 plain (256 channels: a 128 + 128 concat) or 24 split (3 per 32 channels)

struct SrcDev {
    const char* data;
    int C;              // channels of this tensor
    int up;             // 1: half resolution, read at (x>>1, y>>1, z>>1)
    int Ys, Zs;         // its own y / z extents
    long long plane;    // bytes per x-plane
    long long batch;    // bytes per batch item
};

struct Conv3Args {
    SrcDev src[2];
    int nchunks, c0chunks;
    const char* wpk;
    const float* bias;
    char* out;
    float* partial;
    const char* zeros;
    int B, Xt, Yt, Zt;
    int XC, nxc, npatch;
    int mode;           // 0: linear (y,z) ranges (small Zt), 1: TY x TZ rectangles
    int TZ, nzc;
    int pitch, nposp;
    int w8_off, w8_scale, wpk_bytes;   // MIX8: byte offset of the fp8 fragments in wpk, E8M0 scale word of the weights, bytes of wpk
    int jstep;          // conv3_m16_kernel: region positions from a lane's voxel c16 to voxel 16 + c16 of its column tile (16 | pitch)
    // act[i] != NULL: source i is the RAW output of its producing conv and act[i] (B, 2, C_i) the affine of its GroupNorm:
    // every lane applies silu(a*x + b) to the chunks it staged itself, in LDS, once its own LDS-DMA has landed -- the
    // fused form of GroupNorm + SiLU (no separate pass over the tensor; same arithmetic as gn_silu_kernel, bit-identical)
    const float* act[2];
    long long* dbg;     // -DSK_TIMING builds: per-wave phase cycle sums
    int dbg_off;        // ... of the workgroups blockIdx.x in [dbg_off, dbg_off + 4096) (SK_CONV_DBG_OFF)
    int alt;            // multi-chunk layers: visit the chunks in alternating order (see `reuse` in the kernels)
    int has_box, box_lo[3], box_hi[3];   // sk_conv3d_box: only the output voxels inside [lo, hi) are STORED (conv3_px_kernel)
    int ablate;         // timing experiments only (-DSK_TUNING builds, SK_CONV_ABLATE): 1 skip DMA, 2 reuse first weights, 4 skip stores
    // per phase chunk: bit 0 = source, bit 1 = "same LDS image as the previous chunk: no DMA", bits 8.. = byte offset of
    // the chunk inside the source's voxel line
    unsigned chinfo[kMaxChunks];
};

// Timing experiments (wrong results by design) exist only in the -DSK_TUNING build that tools/ use: in the release
// library no environment variable can change what a kernel computes.
// (non-temporal epilogue stores: -1.7 % on the conv kernels, +4.5 % on the GroupNorm pass that reads the tensor next --
// it loses what the conv's stores leave in the Infinity Cache; net zero, not used)
// Phase timing (tools/conv_phase_timing.py, -DSK_TIMING build only): per wave, cycles between the marks of a phase
#ifdef SK_TIMING
#define SK_T_DECL long long tacc_[sk::kTimingSlots] = {0}; long long tprev_ = __builtin_readcyclecounter();
#define SK_T(i) { const long long t_ = __builtin_readcyclecounter(); tacc_[i] += t_ - tprev_; tprev_ = t_; }
#define SK_T_DUMP(a, w, lane) if ((a).dbg && (int)blockIdx.x >= (a).dbg_off && (int)blockIdx.x < (a).dbg_off + sk::kTimingBlocks && (w) < 4 && (lane) == 0) { \
        for (int i_ = 0; i_ < sk::kTimingSlots; ++i_) (a).dbg[((long long)((int)blockIdx.x - (a).dbg_off) * 4 + (w)) * sk::kTimingSlots + i_] = tacc_[i_]; }
#else
#define SK_T_DECL
#define SK_T(i)
#define SK_T_DUMP(a, w, lane)
#endif
#ifdef SK_TUNING
#define SK_ABL(a, bits) ((a).ablate & (bits))
#else
#define SK_ABL(a, bits) 0
#endif
// compile-time ablations of conv3_px_kernel's MFMA body (a run-time switch there changes the schedule it is meant to
// measure): make ... EXTRA="-DSK_TUNING -DSK_PX_ABLATE=48"; 16: no LDS weight reads, 32: no B fragment reads after a
// step's first tap row, 64: no barrier between the steps.  Results are wrong by design.
#ifndef SK_PX_ABLATE
#define SK_PX_ABLATE 0
#endif
#define SK_PX_ABL(bits) ((SK_PX_ABLATE) & (bits))

__device__ __forceinline__ void dma16(const char* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Wave layout inside a workgroup: NT = COUT/32 cout tiles.  The 4 waves form NT groups
// along cout x (4/NT) groups along the voxel columns; a wave owns ONE cout tile and
// P = NT column tiles (32 voxels each) of the 128-voxel patch.  So a wave streams only its
// own cout tile's weights (weight-fragment reuse = XS*P MFMAs per 1 KiB fragment) and the
// activation fragment of a column tile is read once per wave that needs it.
// RES: rows (dy,dz) of the 9 whose weight fragments stay in registers for the whole workgroup -- only for
// single-chunk layers (the same 54 fragments every step) of COUT 32, where the kernel leaves ~80 of its 256
// registers: 3 rows = 18 fragments = 72 VGPRs (246 in all; 4 rows spill), the other 6 rows stream as before.
// (Keeping chunk 0's rows resident in a two-chunk layer, selected at run time per phase, spills: 1.8x slower;
// requesting a whole row of streamed fragments two bodies ahead instead of one measured 4 % slower.)
// Measured A/B on one device: -4..5 % time on enc0.1 / dec0.1.
// MIX8 (precision "mix8", SPLIT tensors): the phases of a step alternate between a logical chunk's fp16 product (its hi
// halves against w_hi) and ONE block-scaled fp8 product over K = 64 = [x8 (32) | lo8 (32)] . [w_lo | w] per tap
// (v_mfma_scale_f32_32x32x64_f8f6f4: lanes 0-31 hold K 0-31, lanes 32-63 K 32-63 of their row / column) -- see conv3_m16_kernel.
template <int COUT, int XS, int RES = 0, bool SPLIT = false, bool MIX8 = false>
__global__ void __launch_bounds__(256, 2) conv3_kernel(Conv3Args a) {
    constexpr int NT = COUT / 32;
    constexpr int P = NT;            // column tiles per wave
    constexpr int R = XS + 2;
    constexpr bool kLateWait = (NT != 2);  // wait for the next phase's DMA after the epilogue (see there)
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w % NT;           // cout tile of this wave
    const int wm = w / NT;           // voxel group of this wave
    const int col = lane & 31, h = lane >> 5;

    // Workgroups are dealt round-robin over the 8 XCDs (private L2 each).  Neighbouring patches
    // share their y/z halo and consecutive x-chunks share two planes: give each XCD a contiguous
    // run of the (batch, x-chunk, patch) order so those re-reads hit its own L2 (bijective remap).
    int blk = blockIdx.x;
    if (!SK_ABL(a, 16)) {
        const int nwg = gridDim.x, xcd = blk & 7, qn = nwg >> 3, rn = nwg & 7;
        blk = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blk >> 3);
    }
    const int patch = blk % a.npatch;
    blk /= a.npatch;
    const int xc = blk % a.nxc;
    const int b = blk / a.nxc;
    const int block_in_batch = xc * a.npatch + patch;
    const int nblk = a.npatch * a.nxc;

    // ---- patch geometry ------------------------------------------------------------
    const int pitch = a.pitch;
    int off, ybase, zbase;      // region position q -> (y, z): Pq = q + off; y = ybase + Pq/pitch; z = zbase + Pq%pitch
    int q_row[P];               // region position of this lane's voxel in column tile p
    int out_vox0[P];            // in-plane voxel index of column 0 of tile p (the 32 columns are contiguous)
    int tile_nvox[P];           // columns with out_vox0 + c < tile_nvox are inside the tile
    bool vvalid[P];
    bool zlo[P], zhi[P];        // linear mode: this lane's voxel sits on the z = 0 / z = Zt-1 face
    if (a.mode == 0) {
        // Linear mode: region position q <-> in-plane voxel index v0 - Zt - 1 + q, NO z halo (pitch = Zt), so
        // the 32 columns of a tile are 32 consecutive positions for every tap: with the 16-byte chunk swizzle
        // below any 16 consecutive positions cover all 64 banks, i.e. every ds_read_b128 lane group is
        // conflict-free (a z halo makes q jump by 2 at each row end: SQ_LDS_BANK_CONFLICT was ~50 % of the
        // LDS cycles).  A dz = -1 / +1 tap on the z = 0 / Zt-1 face would read the neighbouring row's voxel:
        // those lanes read the plane's zero position instead (one v_cndmask on the address).
        int v0 = patch * kPatch;
        off = v0 - a.Zt - 1;
        ybase = 0;
        zbase = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            int v = v0 + 32 * (wm * P + p) + col;
            int vy = v / a.Zt, vz = v - vy * a.Zt;
            vvalid[p] = v < a.Yt * a.Zt;
            zlo[p] = vz == 0;
            zhi[p] = vz == a.Zt - 1;
            q_row[p] = v - off;
            out_vox0[p] = v0 + 32 * (wm * P + p);
            tile_nvox[p] = a.Yt * a.Zt;
        }
    } else {
        const int TY = kPatch / a.TZ;
        int yg = patch / a.nzc, zc = patch - yg * a.nzc;
        int y0 = yg * TY, zc0 = zc * a.TZ;
        off = 0;
        ybase = y0 - 1;
        zbase = zc0 - 1;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            int vl = 32 * (wm * P + p) + col;
            int yl = vl / a.TZ, zl = vl - yl * a.TZ;
            int vy = y0 + yl, vz = zc0 + zl;
            vvalid[p] = vy < a.Yt && vz < a.Zt;
            zlo[p] = zhi[p] = false;
            q_row[p] = (yl + 1) * pitch + (zl + 1);
            // a column tile is one 32-voxel z segment of one line (TZ == 32)
            const int ty = y0 + (32 * (wm * P + p)) / a.TZ;
            out_vox0[p] = ty * a.Zt + zc0;
            tile_nvox[p] = ty < a.Yt ? ty * a.Zt + a.Zt : 0;
        }
    }

    // ---- DMA bookkeeping: this lane's slots of a plane --------------------------------
    const int ndma = a.nposp / 16;  // wave-instructions per plane
    int d_vox[kMaxDma], d_up[kMaxDma];
    // source 16-byte chunk (swizzle on the SOURCE side): slot c of position q holds chunk c ^ ((q >> 2) & 3); q >> 2 =
    // 4 t + (lane >> 4), so the term is the same for every DMA instruction t of this lane
    const int d_cs = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
#pragma unroll
    for (int k = 0; k < kMaxDma; ++k) {
        int t = w + 4 * k;
        int slot = 64 * t + lane;
        int q = slot >> 2, c = slot & 3;
        int Pq = q + off;        // linear mode: the voxel index itself (may be < 0 above the tile)
        int y = ybase + (Pq >= 0 ? Pq / pitch : -1), z = zbase + (Pq >= 0 ? Pq % pitch : 0);
        bool ok = (t < ndma) && y >= 0 && y < a.Yt && z >= 0 && z < a.Zt;
        d_vox[k] = ok ? y * a.Zt + z : -1;
        d_up[k] = ok ? (y >> 1) * a.src[1].Zs + (z >> 1) : -1;
        (void)c;
    }

    // ---- accumulators + GroupNorm partials ----------------------------------------------
    f32x16 acc[P][XS];
    float gsum[4], gsq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) gsum[q] = gsq[q] = 0.0f;

    const int xa = xc * a.XC;
    const int xb = min(xa + a.XC, a.Xt);
    const int plane_bytes = (a.nposp + sk::kZeroPos) * kPosBytes;  // + the zero window (never written by the DMA)
    const int zero_addr = a.nposp * kPosBytes;
    // Plane reuse: when a phase multiplies the SAME chunk as the phase before it, one step further along x, the two
    // trailing planes of that chunk are still staged and only XS new planes are loaded (`reuse`); the slots rotate
    // (`rot` = slot of plane 0).  Single-chunk layers: every phase after the first.  Multi-chunk layers visit their
    // chunks in alternating order (0..n-1, n-1..0, ...: a.alt), so the first phase of every step after the first
    // reuses -- 1/6 (two chunks, XS 4) of the plane loads; holding two planes of EVERY chunk across a step would
    // need (XS + 2) + 2 (n - 1) plane slots, more than 80 KiB at any XS that keeps two workgroups per CU.
    const int nck = a.nchunks;
    // the order is a function of the ABSOLUTE x position of the step (x-chunks start at multiples of XS), so that the
    // summation order of an output plane -- and with it every bit of the result -- does not depend on how the launch
    // was cut into x-chunks (which varies with the batch size)
    auto chunk_of = [&](int step, int k) { return (a.alt && ((xc * (a.XC / XS) + step) & 1)) ? nck - 1 - k : k; };
    const int ch0 = chunk_of(0, 0);
    // SPLIT: the output voxel line is [hi (COUT fp16) | lo (COUT fp16)], value = hi + lo (~22 significant bits)
    constexpr int kOvs = COUT * 2 * (SPLIT ? 2 : 1);   // bytes per output voxel
    const long long out_plane = (long long)a.Yt * a.Zt * kOvs;
    char* outb = a.out + (long long)b * a.Xt * out_plane;

    // bias is the initial accumulator: row (cout) = 32*wn + (r&3) + 8(r>>2) + 4h (re-read per step: 16 registers less in the loop)
    const float* biasp = a.bias + 32 * wn + 4 * h;

    // ---- phase sequence ---------------------------------------------------------------------
    // A phase = (step, chunk): 27 taps of one 32-channel chunk for the XS output planes of a step.
    // After a phase's MFMAs: barrier (LDS planes free) -> LDS-DMA of the NEXT phase's planes ->
    // wait for it -> if the step is complete, its epilogue (transpose + stores + GroupNorm
    // partials) -> bare s_barrier.  The epilogue's global stores are never waited for.
    const int nsteps = (xb - xa + XS - 1) / XS;
    const int nphases = nsteps * a.nchunks;
    char* pad = lds + R * plane_bytes + w * kPadBytes;
    const int rv = lane >> 2, rc = lane & 3;  // epilogue read-back: voxel (0..15), 16-byte chunk

    // buffer-resource LDS-DMA and stores: 32-bit offsets into this batch item's tensors (one address VGPR instead of
    // two, 32-bit address arithmetic: -2 % time); an out-of-range offset reads zeros -- what the halo / padding lanes
    // want -- and drops a masked store
    // A descriptor covers only the x-planes of ONE step (built per phase from wave-uniform values: a few scalar
    // instructions): a tensor of one batch item may exceed the 4 GiB a descriptor / a 32-bit offset can span
    // (the split mode's 512x512x128 tile: 4.3 GB per 32-channel tensor).
    auto issue_dma = [&](int step, int ch, bool reuse, int rot_n) {
        const int x0 = xa + step * XS;
        const unsigned ci = a.chinfo[ch];
        const int si = ci & 1;
        const SrcDev s = a.src[si];
        const int choff = ci >> 8;  // byte offset of the chunk in the voxel line
        const int first_new = reuse ? 2 : 0;  // planes 0,1 are the previous phase's planes XS, XS + 1
        const int xlo = s.up ? (max(x0 - 1, 0) >> 1) : max(x0 - 1, 0);   // first source plane of the step
        const long long wbytes = min((long long)(R + 1) * s.plane, s.batch - (long long)xlo * s.plane);
        const __amdgpu_buffer_rsrc_t rsrc = sk::make_rsrc(s.data + (long long)b * s.batch + (long long)xlo * s.plane, (unsigned)wbytes);
        for (int i = SK_ABL(a, 1) ? R : first_new; i < R; ++i) {
            const int x = x0 - 1 + i;
            const int slotp = (rot_n + i) % R;
            const bool xok = x >= 0 && x < a.Xt;
            char* lbase = lds + slotp * plane_bytes;
            const unsigned xoff = (unsigned)(((s.up ? (x >> 1) : x) - xlo) * (int)s.plane + choff + d_cs);
            const unsigned vstride = (unsigned)(s.C * 2);
#pragma unroll
            for (int k = 0; k < kMaxDma; ++k) {
                const int t = w + 4 * k;
                if (t < ndma) {
                    const int vox = s.up ? d_up[k] : d_vox[k];
                    const unsigned voff = (xok && vox >= 0) ? xoff + (unsigned)vox * vstride : sk::kOob;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lbase + t * 1024), 16, voff, 0, 0, 0);
                }
            }
        }
    };
    // weight fragment index: (((ch*9 + dydz)*2 + ks)*3 + d)*NT + nt
    // weight fragments through a buffer resource: wave-uniform byte offset in an SGPR + the constant lane * 16 in one VGPR
    // (no 64-bit address arithmetic in the tap loop)
    auto wbase = [&](int ch) { return (unsigned)(((MIX8 ? ch >> 1 : ch) * (9 * 2 * 3 * NT) + wn) * 1024); };   // MIX8: phases 2 k, 2 k + 1 = chunk k
    const __amdgpu_buffer_rsrc_t wrsrc = sk::make_rsrc(a.wpk, MIX8 ? (unsigned)a.wpk_bytes : (unsigned)(a.nchunks * (9 * 2 * 3 * NT) * 1024));
    const unsigned wlane = lane * 16;
    auto wload = [&](unsigned off) {
        return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, __builtin_amdgcn_readfirstlane(off), 0));
    };

    half8 a0[3], a1[3];
    half8 wres[RES > 0 ? 2 * RES : 1][3];
    if constexpr (RES > 0) {
#pragma unroll
        for (int r = 0; r < 2 * RES; ++r)
#pragma unroll
            for (int d = 0; d < 3; ++d) wres[r][d] = wload(wbase(0) + ((r * 3 + d) * NT) * 1024);
    }
    if (tid < R * 4 * sk::kZeroPos)
        *reinterpret_cast<uint4*>(lds + (tid / (4 * sk::kZeroPos)) * plane_bytes + zero_addr + (tid % (4 * sk::kZeroPos)) * 16) = make_uint4(0, 0, 0, 0);
    issue_dma(0, ch0, false, 0);
    if constexpr (RES == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) a0[d] = wload(wbase(ch0) + (d * NT) * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int step = 0, k = 0, ch = ch0, rot = 0;   // k: position of the phase in its step's chunk order
    SK_T_DECL
    // MIX8: the phases of a step strictly alternate (fp16, fp8): the loop runs over pairs with the kind a compile-time value --
    // a run-time branch between two bodies that both own all 128 accumulator registers costs copies and spills at the joins
    constexpr int kSub = MIX8 ? 2 : 1;
    for (int ph0 = 0; ph0 < nphases; ph0 += kSub)
#pragma unroll
    for (int sub = 0; sub < kSub; ++sub) {
        const int ph = ph0 + sub;
        const int x0 = xa + step * XS;
        if (k == 0) {
            f32x16 binit;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(biasp + 8 * q);
                binit[4 * q + 0] = bv[0];
                binit[4 * q + 1] = bv[1];
                binit[4 * q + 2] = bv[2];
                binit[4 * q + 3] = bv[3];
            }
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int o = 0; o < XS; ++o) acc[p][o] = binit;
        }
        // ---------------- MFMA over the 27 taps of this chunk ---------------------------
        {
            const unsigned wch = wbase(ch);
            int pslot[R];
#pragma unroll
            for (int i = 0; i < R; ++i) pslot[i] = ((rot + i) % R) * plane_bytes;

            auto compute = [&](int dydz, int ks, const half8 (&afr)[3]) {
                const int dz = dydz % 3 - 1;
                const int tapoff = (dydz / 3 - 1) * pitch + dz;
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int q = q_row[p] + tapoff;
                    int addr = (q * 4 + ((ks * 2 + h) ^ ((q >> 2) & 3))) * 16;
                    if (dz < 0) addr = zlo[p] ? sk::zero_of(zero_addr, addr) : addr;   // COUT 64 (this lambda's kernel variant)
                    if (dz > 0) addr = zhi[p] ? sk::zero_of(zero_addr, addr) : addr;
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const half8 bfr = *reinterpret_cast<const half8*>(lds + pslot[i] + addr);
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            const int o = i - d;  // x_in = x_out + (d - 1)
                            if (o >= 0 && o < XS)
                                acc[p][o] = SK_MFMA_32x32x16_T16(afr[d], bfr, acc[p][o], 0, 0, 0);
                        }
                    }
                }
            };
            // hipcc schedules the ds_read / MFMA interleave of a (dydz, ks) body itself (pinning
            // it with sched_barrier measured 8 % slower); the weight fragments of the next body
            // are requested one body ahead.
            // Fully unrolling the 9 tap rows lets hipcc hoist the next row's loads: +1.5 % for
            // COUT 32 / 64, -8 % for COUT 128 (code size), measured A/B on one device.
            // The machine scheduler sinks the weight-fragment loads towards their first use (64-100 cycles of
            // latency hiding left in the ISA): a scheduling barrier after each load group keeps them a whole body
            // (24 MFMAs) ahead -- -3.7 % time on COUT 64 and 128 (tools/kernel_ab.sh, round 2); the 16x16x32 kernel
            // below loses 4 % with the same barrier and keeps the free schedule.
            constexpr int kTapUnroll = (NT <= 2) ? 9 : 1;
            const bool f8phase = MIX8 && sub == 1;   // (= a.chinfo[ch] & 4: the host lists a chunk's phases in this order, no `alt`)
            if (f8phase) {
                if constexpr (MIX8) {
                    typedef int v8i __attribute__((ext_vector_type(8)));
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const int sa = a.w8_scale, sb = 0x70707070;   // E8M0: weights 2^-b (host), activations 2^-15
                    // fp8 fragment ((chunk, dydz, d), cout tile): 2 KiB, lane l = 32 bytes: row l & 31, K block l >> 5
                    const unsigned w8 = (unsigned)a.w8_off + (unsigned)(((ch >> 1) * 27) * NT + wn) * 2048u + wlane * 2;
#pragma unroll 1
                    for (int dydz = 0; dydz < 9; ++dydz) {
                        const int dz = dydz % 3 - 1;
                        const int tapoff = (dydz / 3 - 1) * pitch + dz;
                        v8i af[3];
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            const unsigned wo = w8 + (unsigned)((dydz * 3 + d) * NT) * 2048u;
                            const u32x4 w0 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wo, 0, 0);
                            const u32x4 w1 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wo + 16, 0, 0);
                            af[d] = v8i{(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
                        }
#pragma unroll
                        for (int p = 0; p < P; ++p) {
                            const int q = q_row[p] + tapoff;
                            // this lane's K block: h = 0 the x8 bytes (chunks 0, 1 of the staged 64), h = 1 the lo8 bytes (chunks 2, 3)
                            int addr0 = (q * 4 + ((2 * h) ^ ((q >> 2) & 3))) * 16;
                            int addr1 = (q * 4 + ((2 * h + 1) ^ ((q >> 2) & 3))) * 16;
                            if ((dz < 0 && zlo[p]) || (dz > 0 && zhi[p])) {
                                addr0 = sk::zero_of(zero_addr, addr0);
                                addr1 = sk::zero_of(zero_addr, addr1);
                            }
#pragma unroll
                            for (int i = 0; i < R; ++i) {
                                const u32x4 lo4 = *reinterpret_cast<const u32x4*>(lds + pslot[i] + addr0);
                                const u32x4 hi4 = *reinterpret_cast<const u32x4*>(lds + pslot[i] + addr1);
                                const v8i bf = v8i{(int)lo4[0], (int)lo4[1], (int)lo4[2], (int)lo4[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
#pragma unroll
                                for (int d = 0; d < 3; ++d) {
                                    const int o = i - d;
                                    if (o >= 0 && o < XS)
                                        acc[p][o] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[d], bf, acc[p][o], 0, 0, 0, sa, 0, sb);
                                }
                            }
                        }
                    }
                }
            } else if constexpr (NT == 4) {
            // COUT 128 (measured: -6 % on the 128 -> 128 layers, nothing on COUT 64, tools/layer_ab.py) -- B fragments one body ahead: a body = (dydz, ks, p) = R fragments (one per staged plane) feeding 3 XS MFMAs.
            // The compiler's own schedule requests a fragment one or two MFMAs before its first use (`ds_read ;
            // s_waitcnt lgkmcnt(1) ; v_mfma` all along the tap loop): a wave that has the SIMD to itself then waits out
            // most of every LDS latency.  Here body n + 1's fragments are requested before body n's MFMAs.
            half8 bq[2][R];
            auto bload = [&](int dydz, int ks, int p, half8 (&dst)[R]) {
                const int dz = dydz % 3 - 1;
                const int tapoff = (dydz / 3 - 1) * pitch + dz;
                const int q = q_row[p] + tapoff;
                int addr = (q * 4 + ((ks * 2 + h) ^ ((q >> 2) & 3))) * 16;
                // (COUT 128, this path: the shared zero line -- the bank-matched window measured +2.4 % time here although it
                // takes the conflict share from 0.35 to 0.03: profiles/r04_conv_sq_counters_zero_window_*.json)
                if (dz < 0) addr = zlo[p] ? zero_addr : addr;
                if (dz > 0) addr = zhi[p] ? zero_addr : addr;
#pragma unroll
                for (int i = 0; i < R; ++i) dst[i] = *reinterpret_cast<const half8*>(lds + pslot[i] + addr);
            };
            auto bmma = [&](int p, const half8 (&afr)[3], const half8 (&src)[R]) {
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        const int o = i - d;  // x_in = x_out + (d - 1)
                        if (o >= 0 && o < XS) acc[p][o] = SK_MFMA_32x32x16_T16(afr[d], src[i], acc[p][o], 0, 0, 0);
                    }
            };
            bload(0, 0, 0, bq[0]);
#pragma unroll kTapUnroll
            for (int dydz = 0; dydz < 9; ++dydz) {
                const unsigned wrow = wch + (unsigned)(dydz * 2) * (3 * NT) * 1024;
#pragma unroll
                for (int d = 0; d < 3; ++d) a1[d] = wload(wrow + ((3 + d) * NT) * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const int n = ks * P + p;   // a row has 2 P bodies: the buffer of a body is a compile-time index
                        int nd = dydz, nks = ks, np = p + 1;
                        if (np == P) {
                            np = 0;
                            ++nks;
                        }
                        if (nks == 2) {
                            nks = 0;
                            ++nd;
                        }
                        if (nd < 9) bload(nd, nks, np, bq[(n + 1) & 1]);
                        // one LDS read of the next body per MFMA gap of this one, the rest of the MFMAs behind them
                        // (all R reads in one burst ahead of the MFMAs: +2 % instead)
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 3 * XS - R, 0);
                        if (ks == 0)
                            bmma(p, a0, bq[n & 1]);
                        else
                            bmma(p, a1, bq[n & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (ks == 0 && dydz < 8) {
#pragma unroll
                        for (int d = 0; d < 3; ++d) a0[d] = wload(wrow + ((6 + d) * NT) * 1024);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            } else
#pragma unroll kTapUnroll
            for (int dydz = 0; dydz < 9; ++dydz) {
                const unsigned wrow = wch + (unsigned)((SK_ABL(a, 2) ? 0 : dydz) * 2) * (3 * NT) * 1024;
                if constexpr (RES > 0) {
                    if (dydz < RES) {
                        compute(dydz, 0, wres[2 * dydz]);
                        if (dydz == RES - 1) {  // first streamed row's ks = 0 fragments, one body ahead
#pragma unroll
                            for (int d = 0; d < 3; ++d)
                                a0[d] = wload(wrow + ((6 + d) * NT) * 1024);
                        }
                        compute(dydz, 1, wres[2 * dydz + 1]);
                        continue;
                    }
                }
                if (!SK_ABL(a, 32) || dydz == 0) {
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                        a1[d] = wload(wrow + ((3 + d) * NT) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);   // see kTapUnroll's comment
                compute(dydz, 0, a0);
                if (dydz < 8 && !SK_ABL(a, 32)) {
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                        a0[d] = wload(wrow + ((6 + d) * NT) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);   // see kTapUnroll's comment
                compute(dydz, 1, a1);
            }
        }

        // ---------------- hand the LDS planes to the next phase ---------------------------
        const bool step_done = (k == nck - 1);
        int nstep = step, nk = k + 1;
        if (step_done) {
            nstep = step + 1;
            nk = 0;
        }
        const int nch = chunk_of(nstep, nk);
        const bool have_next = ph + 1 < nphases;
        const bool reuse_n = step_done && nch == ch;   // same chunk, next step: its planes XS, XS + 1 stay
        const int rot_n = reuse_n ? (rot + XS) % R : rot;
        SK_T(0)           // MFMA phase
        __syncthreads();  // every wave is done reading the planes about to be overwritten
        SK_T(1)           // barrier: waiting for the slowest wave of the workgroup
        if (have_next) {
            if (!(a.chinfo[nch] & 2)) issue_dma(nstep, nch, reuse_n, rot_n);   // bit 1: the next chunk multiplies the SAME staged planes
            if constexpr (RES == 0) {
#pragma unroll
                for (int d = 0; d < 3; ++d) a0[d] = wload(wbase(nch) + (d * NT) * 1024);
            }
            // The DMA must have landed before this wave passes the closing barrier.  vmcnt retires in issue
            // order, so after an epilogue that issues a FIXED number of stores (invalid lanes / planes store to
            // a scratch line instead of branching) `vmcnt(that number)` means "everything older than the
            // stores -- the DMA and the weight fragments -- has landed": the DMA latency hides behind the
            // epilogue's transposes and the stores still drain under the next phase's MFMAs.
            // Measured per layer (same device, A/B): -4 % time for COUT 32, +4 % for COUT 64 (two column tiles per
            // wave: a longer epilogue), neutral for COUT 128 -> late wait for NT != 2 only.
            if (!kLateWait || !step_done || SK_ABL(a, ~0)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }

        SK_T(2)           // LDS-DMA issue + weight prefetch (+ the landing wait where it is not deferred)
        // ---------------- epilogue (overlaps the DMA): raw fp16 store + GroupNorm partials ------
        // The accumulator tile is [cout rows in registers][voxel columns on lanes]; the output
        // is channels-last.  Each wave transposes its 32x32 tile through a private 2 KiB LDS
        // pad (16-B chunks XOR-swizzled) so that every lane stores 16 contiguous bytes and one
        // store instruction writes 1 KiB of whole 64-B voxel lines (4 lanes per line).
        if (step_done) {
            char* outw = outb + (long long)x0 * out_plane;   // the step's XS output planes: one descriptor
            const __amdgpu_buffer_rsrc_t rout = sk::make_rsrc(outw, (unsigned)(XS * out_plane));
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int tile_vox0 = out_vox0[p];  // in-plane index of the tile's first voxel
#pragma unroll
                for (int o = 0; o < XS; ++o) {
                    const int x = x0 + o;
                    const bool ok = vvalid[p] && x < xb;
#pragma unroll
                    for (int part = 0; part < (SPLIT ? 2 : 1); ++part) {   // SPLIT: the hi halves, then the lo halves
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float v0 = acc[p][o][4 * q], v1 = acc[p][o][4 * q + 1];
                            float v2 = acc[p][o][4 * q + 2], v3 = acc[p][o][4 * q + 3];
                            half4 hv = {(t16)v0, (t16)v1, (t16)v2, (t16)v3};
                            if (part == 1)   // lo = fp16(v - hi): exact difference, rounded once
                                hv = half4{(t16)(v0 - (float)hv[0]), (t16)(v1 - (float)hv[1]),
                                           (t16)(v2 - (float)hv[2]), (t16)(v3 - (float)hv[3])};
                            *reinterpret_cast<half4*>(pad + col * kPadStride + ((q ^ ((col >> 1) & 3)) * 16) + 8 * h) = hv;
                            if (SPLIT) {
                                if (ok && part == 0) {   // split: statistics of the fp32 accumulators
                                    gsum[q] += (v0 + v1) + (v2 + v3);
                                    gsq[q] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
                                }
                            } else {   // statistics of the stored 16-bit values on v_dot2c (see conv3_m16_kernel)
                                const t16x2 z2 = {(t16)0.0f, (t16)0.0f}, one2 = {(t16)1.0f, (t16)1.0f};
                                const t16x2 lo2 = ok ? t16x2{hv[0], hv[1]} : z2, hi2 = ok ? t16x2{hv[2], hv[3]} : z2;
                                gsum[q] = SK_DOT2_T16(lo2, one2, gsum[q]);
                                gsum[q] = SK_DOT2_T16(hi2, one2, gsum[q]);
                                gsq[q] = SK_DOT2_T16(lo2, lo2, gsq[q]);
                                gsq[q] = SK_DOT2_T16(hi2, hi2, gsq[q]);
                            }
                        }
                        if (!SK_ABL(a, 4)) {
                            char* op = outb + (long long)x * out_plane + (long long)tile_vox0 * kOvs + wn * 64 + part * (COUT * 2);
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh) {
                                const int vv = rv + 16 * hh;
                                const half8 line = *reinterpret_cast<const half8*>(
                                    pad + vv * kPadStride + ((rc ^ ((vv >> 1) & 3)) * 16));
                                const bool sok = x < xb && tile_vox0 + vv < tile_nvox[p];
                                // always issued (the counted wait relies on it); a masked lane's offset is out of range: dropped
                                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, line), rout,
                                                                       sok ? (unsigned)(op + (long long)vv * kOvs + rc * 16 - outw) : sk::kOob, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        SK_T(3)           // epilogue
        if (have_next) {
            if (kLateWait && step_done && !SK_ABL(a, ~0)) {
                constexpr int kStores = P * XS * 2 * (SPLIT ? 2 : 1);  // global stores the epilogue just issued
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kStores) : "memory");
            }
            SK_T(4)       // deferred landing wait
            SK_T(5)       // in-LDS activation
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            SK_T(6)       // closing barrier
        }
        step = nstep;
        k = nk;
        ch = nch;
        rot = rot_n;
    }

    SK_T_DUMP(a, w, lane)
    // ---- block-level reduction of the GroupNorm partials ------------------------------------
    if (a.partial) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);  // [4 waves][8 quads of the wave's cout tile][2]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float s = gsum[q], ss = gsq[q];
#pragma unroll
            for (int m = 16; m > 0; m >>= 1) {
                s += __shfl_xor(s, m);
                ss += __shfl_xor(ss, m);
            }
            if (col == 0) {
                red[(w * 8 + 2 * q + h) * 2 + 0] = s;
                red[(w * 8 + 2 * q + h) * 2 + 1] = ss;
            }
        }
        __syncthreads();
        if (tid < NT * 16) {
            // channel quad Q = cout/4 = 8*nt + k ; waves with wn == nt: w = wm*NT + nt
            const int nt = tid / 16, k2 = tid % 16;
            float t = 0.0f;
#pragma unroll
            for (int g = 0; g < 4 / NT; ++g) t += red[(g * NT + nt) * 16 + k2];
            a.partial[((long long)b * nblk + block_in_batch) * (NT * 16) + tid] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------
// COUT 32 runs on v_mfma_f32_16x16x32_f16 (conv3_m16_kernel): tools/mfma_shape_probe.hip sustains 1.9 PFLOP/s with
// that shape against 1.65 with 32x32x16 on this part, and the COUT-32 variant has the registers for it (a 32 x 32
// tile becomes 2 x 2 results of 16 x 16; K = 32 = the whole channel chunk per instruction; the 16-byte chunk
// swizzle of the staged planes and the fragment packing differ, see below).  A/B on one device: -5 % time on each of
// the three COUT-32 layers (55 % of the conv time).  With this shape the COUT 64 / 128 variants spill, or lose
// reuse when XS is cut to fit (COUT 64: +22 % with a 76-byte spill at XS 4, +6 % at XS 3; COUT 128: +15 %): they stay
// on conv3_kernel (32x32x16) above.
//
// RES: rows (dy,dz) of the 9 whose weight fragments stay in registers for the whole workgroup -- only for
// single-chunk layers (the same 54 fragments every step) of COUT 32, where the kernel leaves ~80 of its 256
// registers: 3 rows = 18 fragments = 72 VGPRs (246 in all; 4 rows spill), the other 6 rows stream as before.
// (Keeping chunk 0's rows resident in a two-chunk layer, selected at run time per phase, spills: 1.8x slower;
// requesting a whole row of streamed fragments two bodies ahead instead of one measured 4 % slower.)
// Measured A/B on one device: -4..5 % time on enc0.1 / dec0.1.
// WL: further tap rows (RES .. RES + WL - 1) whose fragments sit in LDS behind the ring, one copy per workgroup -- the four
// waves of a COUT-32 workgroup stream the SAME fragments, each through the CU's vector-memory path, which these kernels
// load as heavily as the matrix pipe.  Two rows fit next to the six-plane ring of the production
// tile (80 256 B: still two workgroups per CU): 12 of the 36 weight loads per wave and step become LDS reads,
// enc0.1 0.677 -> 0.655 ms, dec0.1 0.867 -> 0.848 ms per 8 tiles (tools/layer_ab.py).
// MIX8 (round 4, precision "mix8"; needs SPLIT for the output format): a logical chunk is an fp16 phase (x_hi w_hi) and ONE fp8 phase
// that carries both cross terms -- K = [x8 | x_lo8] . [w_lo8 | w8] of TWO tap rows per v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3
// operands at twice the fp16 rate, nine tap rows as five pairs) -- instead of split's three fp16 phases.  The source voxel line is
// [hi fp16 (32) | x8 (32) | lo8 (32)], 128 bytes as a split line, so staging is unchanged; see sk_conv3d_mix8.
template <int COUT, int XS, int RES = 0, bool SPLIT = false, int WL = 0, bool MIX8 = false>
__global__ void __launch_bounds__(256, 2) conv3_m16_kernel(Conv3Args a) {
    constexpr int NT = COUT / 32;
    constexpr int P = NT;            // column tiles per wave
    constexpr int R = XS + 2;
    constexpr bool kLateWait = (NT != 2);  // wait for the next phase's DMA after the epilogue (see there)
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w % NT;           // cout tile of this wave
    const int wm = w / NT;           // voxel group of this wave
    // v_mfma_f32_16x16x32_f16 operand roles: c16 = row of A (cout) / column of B (voxel), g = 8-wide K group;
    // a 32 x 32 (cout, voxel) tile is 2 x 2 of its 16 x 16 results: i = cout half, j = voxel half.
    const int c16 = lane & 15, g = lane >> 4;

    // Workgroups are dealt round-robin over the 8 XCDs (private L2 each).  Neighbouring patches
    // share their y/z halo and consecutive x-chunks share two planes: give each XCD a contiguous
    // run of the (batch, x-chunk, patch) order so those re-reads hit its own L2 (bijective remap).
    int blk = blockIdx.x;
    if (!SK_ABL(a, 16)) {
        const int nwg = gridDim.x, xcd = blk & 7, qn = nwg >> 3, rn = nwg & 7;
        blk = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (blk >> 3);
    }
    const int patch = blk % a.npatch;
    blk /= a.npatch;
    const int xc = blk % a.nxc;
    const int b = blk / a.nxc;
    const int block_in_batch = xc * a.npatch + patch;
    const int nblk = a.npatch * a.nxc;

    // ---- patch geometry ------------------------------------------------------------
    const int pitch = a.pitch;
    int off, ybase, zbase;      // region position q -> (y, z): Pq = q + off; y = ybase + Pq/pitch; z = zbase + Pq%pitch
    int q_row[P];               // region position of this lane's voxel c16 of column tile p (voxel 16 + c16: + a.jstep)
    int svox[P];                // in-plane index of the voxel this lane STORES for column tile p (column c16 + 16 (g & 1)
                                // after the permlane transpose), -1: outside the tile
    // per-lane flags of voxel (p, j), bit 2p + j: on the z = 0 face (linear mode) | << 8: on the z = Zt-1 face | << 16:
    // inside the tile.  One VGPR instead of 3 x 2P lane masks (the COUT 128 variant spilled on those).
    unsigned vflags = 0;
    auto zlo = [&](int p, int j) { return (vflags >> (2 * p + j)) & 1u; };
    auto zhi = [&](int p, int j) { return (vflags >> (8 + 2 * p + j)) & 1u; };
    auto vvalid = [&](int p, int j) { return (vflags >> (16 + 2 * p + j)) & 1u; };
    if (a.mode == 0) {
        // Linear mode: region position q <-> in-plane voxel index v0 - Zt - 1 + q, NO z halo (pitch = Zt), so
        // the 32 columns of a tile are 32 consecutive positions for every tap: with the 16-byte chunk swizzle
        // below any 16 consecutive positions cover all 64 banks, i.e. every ds_read_b128 lane group is
        // conflict-free (a z halo makes q jump by 2 at each row end: SQ_LDS_BANK_CONFLICT was ~50 % of the
        // LDS cycles).  A dz = -1 / +1 tap on the z = 0 / Zt-1 face would read the neighbouring row's voxel:
        // those lanes read the plane's zero position instead (one v_cndmask on the address).
        int v0 = patch * kPatch;
        off = v0 - a.Zt - 1;
        ybase = 0;
        zbase = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int v = v0 + 32 * (wm * P + p) + 16 * j + c16;
                int vy = v / a.Zt, vz = v - vy * a.Zt;
                vflags |= (unsigned)(v < a.Yt * a.Zt) << (16 + 2 * p + j);
                vflags |= (unsigned)(vz == 0) << (2 * p + j);
                vflags |= (unsigned)(vz == a.Zt - 1) << (8 + 2 * p + j);
                if (j == 0) q_row[p] = v - off;
            }
            const int sv = v0 + 32 * (wm * P + p) + c16 + 16 * (g & 1);
            svox[p] = sv < a.Yt * a.Zt ? sv : -1;
        }
    } else {
        const int TY = kPatch / a.TZ;
        int yg = patch / a.nzc, zc = patch - yg * a.nzc;
        int y0 = yg * TY, zc0 = zc * a.TZ;
        off = 0;
        ybase = y0 - 1;
        zbase = zc0 - 1;
#pragma unroll
        for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int vl = 32 * (wm * P + p) + 16 * j + c16;
                int yl = vl / a.TZ, zl = vl - yl * a.TZ;
                int vy = y0 + yl, vz = zc0 + zl;
                vflags |= (unsigned)(vy < a.Yt && vz < a.Zt) << (16 + 2 * p + j);
                if (j == 0) q_row[p] = (yl + 1) * pitch + (zl + 1);
            }
            // a column tile is one 32-voxel z segment of one line (TZ = 32) or two 16-voxel segments of two lines (TZ = 16)
            const int svl = 32 * (wm * P + p) + c16 + 16 * (g & 1);
            const int syl = svl / a.TZ, sy = y0 + syl, sz = zc0 + svl - syl * a.TZ;
            svox[p] = (sy < a.Yt && sz < a.Zt) ? sy * a.Zt + sz : -1;
        }
    }

    // Store box (Conv3Args.has_box, sk_conv3d_box / sk_conv3d_box_split): bit p = the voxel this lane STORES for column tile p
    // (column c16 + 16 (g & 1) after the permlane transpose) lies inside the box in (y, z); the x test is wave-uniform
    unsigned sboxm = ~0u;
    if (a.has_box) {
        sboxm = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int sv = max(svox[p], 0);
            const int sy = sv / a.Zt, sz = sv - sy * a.Zt;
            sboxm |= (unsigned)(sy >= a.box_lo[1] && sy < a.box_hi[1] && sz >= a.box_lo[2] && sz < a.box_hi[2]) << p;
        }
    }

    // ---- DMA bookkeeping: this lane's slots of a plane --------------------------------
    const int ndma = a.nposp / 16;  // wave-instructions per plane
    int d_vox[kMaxDma], d_up[kMaxDma];
    // source 16-byte chunk (swizzle on the SOURCE side): slot s of position q holds chunk s ^ 2*((q>>2)&1), which makes
    // every ds_read_b128 lane group of the B reads (16 positions x the wave's 4 K groups) conflict-free for any
    // alignment of the 16 consecutive positions (searched exhaustively).  q >> 2 = 4 t + (lane >> 4): the same for
    // every DMA instruction t of this lane.
    const int d_cs = ((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16;
#pragma unroll
    for (int k = 0; k < kMaxDma; ++k) {
        int t = w + 4 * k;
        int slot = 64 * t + lane;
        int q = slot >> 2, c = slot & 3;
        int Pq = q + off;        // linear mode: the voxel index itself (may be < 0 above the tile)
        int y = ybase + (Pq >= 0 ? Pq / pitch : -1), z = zbase + (Pq >= 0 ? Pq % pitch : 0);
        bool ok = (t < ndma) && y >= 0 && y < a.Yt && z >= 0 && z < a.Zt;
        d_vox[k] = ok ? y * a.Zt + z : -1;
        d_up[k] = ok ? (y >> 1) * a.src[1].Zs + (z >> 1) : -1;
        (void)c;
    }

    // ---- accumulators + GroupNorm partials ----------------------------------------------
    f32x4 acc[P][XS][2][2];   // [i = cout half][j = voxel half]; element r: cout 32 wn + 16 i + 4 g + r, voxel 16 j + c16
    float gsum[2], gsq[2];    // channel quad 4 i + g of the wave's cout tile
#pragma unroll
    for (int q = 0; q < 2; ++q) gsum[q] = gsq[q] = 0.0f;

    const int xa = xc * a.XC;
    const int xb = min(xa + a.XC, a.Xt);
    const int plane_bytes = (a.nposp + sk::kZeroPos) * kPosBytes;  // + the zero window (never written by the DMA)
    const int zero_addr = a.nposp * kPosBytes;
    // Plane reuse: when a phase multiplies the SAME chunk as the phase before it, one step further along x, the two
    // trailing planes of that chunk are still staged and only XS new planes are loaded (`reuse`); the slots rotate
    // (`rot` = slot of plane 0).  Single-chunk layers: every phase after the first.  Multi-chunk layers visit their
    // chunks in alternating order (0..n-1, n-1..0, ...: a.alt), so the first phase of every step after the first
    // reuses -- 1/6 (two chunks, XS 4) of the plane loads; holding two planes of EVERY chunk across a step would
    // need (XS + 2) + 2 (n - 1) plane slots, more than 80 KiB at any XS that keeps two workgroups per CU.
    const int nck = a.nchunks;
    // the order is a function of the ABSOLUTE x position of the step (x-chunks start at multiples of XS), so that the
    // summation order of an output plane -- and with it every bit of the result -- does not depend on how the launch
    // was cut into x-chunks (which varies with the batch size)
    // (MIX8: the two phases of a step -- fp16, fp8 -- swap on odd steps OF THE X-CHUNK, so that a step starts with the kind the
    // step before ended with and reuses its two trailing planes; the x-chunk cut is a function of the tile geometry alone)
    // (SPLIT, one logical chunk = three phases [x_hi w_lo | x_hi w_hi | x_lo w_hi]: odd steps of the x-chunk run them as
    // [x_lo w_hi | x_hi w_lo | x_hi w_hi], so that every step starts on the half the step before ended on)
    auto chunk_of = [&](int step, int k) {
        if (MIX8) return (step & 1) ? nck - 1 - k : k;
        if (SPLIT && nck == 3) return (step & 1) ? (k + 2) % 3 : k;
        return (a.alt && ((xc * (a.XC / XS) + step) & 1)) ? nck - 1 - k : k;
    };
    const int ch0 = chunk_of(0, 0);
    // SPLIT: the output voxel line is [hi (COUT fp16) | lo (COUT fp16)], value = hi + lo (~22 significant bits)
    constexpr int kOvs = COUT * 2 * (SPLIT ? 2 : 1);   // bytes per output voxel
    const long long out_plane = (long long)a.Yt * a.Zt * kOvs;
    char* outb = a.out + (long long)b * a.Xt * out_plane;

    // bias is the initial accumulator: cout = 32 wn + 16 i + 4 g + r (re-read per step: 8 registers less in the loop)
    const float* biasp = a.bias + 32 * wn + 4 * g;

    // ---- phase sequence ---------------------------------------------------------------------
    // A phase = (step, chunk): 27 taps of one 32-channel chunk for the XS output planes of a step.
    // After a phase's MFMAs: barrier (LDS planes free) -> LDS-DMA of the NEXT phase's planes ->
    // wait for it -> if the step is complete, its epilogue (transpose + stores + GroupNorm
    // partials) -> bare s_barrier.  The epilogue's global stores are never waited for.
    const int nsteps = (xb - xa + XS - 1) / XS;
    const int nphases = nsteps * a.nchunks;

    // buffer-resource LDS-DMA and stores (see conv3_kernel)
    // A descriptor covers only the x-planes of ONE step (built per phase from wave-uniform values: a few scalar
    // instructions): a tensor of one batch item may exceed the 4 GiB a descriptor / a 32-bit offset can span
    // (the split mode's 512x512x128 tile: 4.3 GB per 32-channel tensor).
    auto issue_dma = [&](int step, int ch, bool reuse, int rot_n) {
        const int x0 = xa + step * XS;
        const unsigned ci = a.chinfo[ch];
        const int si = ci & 1;
        const SrcDev s = a.src[si];
        const int choff = ci >> 8;  // byte offset of the chunk in the voxel line
        const int first_new = reuse ? 2 : 0;  // planes 0,1 are the previous phase's planes XS, XS + 1
        const int xlo = s.up ? (max(x0 - 1, 0) >> 1) : max(x0 - 1, 0);   // first source plane of the step
        const long long wbytes = min((long long)(R + 1) * s.plane, s.batch - (long long)xlo * s.plane);
        const __amdgpu_buffer_rsrc_t rsrc = sk::make_rsrc(s.data + (long long)b * s.batch + (long long)xlo * s.plane, (unsigned)wbytes);
        for (int i = SK_ABL(a, 1) ? R : first_new; i < R; ++i) {
            const int x = x0 - 1 + i;
            const int slotp = (rot_n + i) % R;
            const bool xok = x >= 0 && x < a.Xt;
            char* lbase = lds + slotp * plane_bytes;
            const unsigned xoff = (unsigned)(((s.up ? (x >> 1) : x) - xlo) * (int)s.plane + choff + d_cs);
            const unsigned vstride = (unsigned)(s.C * 2);
#pragma unroll
            for (int k = 0; k < kMaxDma; ++k) {
                const int t = w + 4 * k;
                if (t < ndma) {
                    const int vox = s.up ? d_up[k] : d_vox[k];
                    const unsigned voff = (xok && vox >= 0) ? xoff + (unsigned)vox * vstride : sk::kOob;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lbase + t * 1024), 16, voff, 0, 0, 0);
                }
            }
        }
    };
    // GroupNorm affine + SiLU of a RAW source, applied in LDS by the lane that staged the chunk (its own `vmcnt` wait
    // covers its own LDS-DMA: no barrier in between); halo / padding lanes staged zeros, which stay zeros
    auto activate = [&](int step, int ch, bool reuse, int rot_n) {
        const unsigned ci = a.chinfo[ch];
        const int si = ci & 1;
        const float* af = a.act[si];
        if (af == nullptr) return;
        const SrcDev s = a.src[si];
        const int x0 = xa + step * XS;
        const int c0 = (int)(ci >> 8) / 2 + (d_cs / 16) * 8;   // first channel (of source si) of this lane's 16-byte chunk
        float ga[8], gb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ga[j] = af[((long long)b * 2) * (s.C) + c0 + j];
            gb[j] = af[((long long)b * 2 + 1) * (s.C) + c0 + j];
        }
        for (int i = reuse ? 2 : 0; i < R; ++i) {
            const int x = x0 - 1 + i;
            if (x < 0 || x >= a.Xt) continue;
            char* lbase = lds + ((rot_n + i) % R) * plane_bytes;
            // the plane's pieces of this lane: all reads first, then branch-free arithmetic (padding lanes keep their
            // zeros through a select, not a divergent branch, so the reads of a plane are in flight together)
            half8 v[kMaxDma];
#pragma unroll
            for (int k = 0; k < kMaxDma; ++k)
                if (w + 4 * k < ndma) v[k] = *reinterpret_cast<const half8*>(lbase + (w + 4 * k) * 1024 + lane * 16);
#pragma unroll
            for (int k = 0; k < kMaxDma; ++k) {
                const int t = w + 4 * k;
                if (t < ndma) {
                    const bool real = (s.up ? d_up[k] : d_vox[k]) >= 0;
                    half8 r;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {   // gn_silu_kernel's arithmetic, op for op
                        const float y = fmaf(ga[e], (float)v[k][e], gb[e]);
                        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-y));
                        const t16 act = sk::round_t16(y * sg);   // unconditional: the asm inside must not sit behind a per-element branch
                        r[e] = real ? act : v[k][e];
                    }
                    *reinterpret_cast<half8*>(lbase + t * 1024 + lane * 16) = r;
                }
            }
        }
    };
    // weight fragment index: (((ch*9 + dydz)*2 + ks)*3 + d)*NT + nt
    // weight fragments through a buffer resource: wave-uniform byte offset in an SGPR + the constant lane * 16 in one VGPR
    // (no 64-bit address arithmetic in the tap loop)
    auto wbase = [&](int ch) { return (unsigned)((ch * (9 * 2 * 3 * NT) + wn) * 1024); };
    const __amdgpu_buffer_rsrc_t wrsrc = sk::make_rsrc(a.wpk, MIX8 ? (unsigned)a.wpk_bytes : (unsigned)(a.nchunks * (9 * 2 * 3 * NT) * 1024));
    const unsigned wlane = lane * 16;
    auto wload = [&](unsigned off) {
        return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, __builtin_amdgcn_readfirstlane(off), 0));
    };

    half8 a0[3], a1[3];
    half8 wres[RES > 0 ? 2 * RES : 1][3];
    if constexpr (RES > 0) {
#pragma unroll
        for (int r = 0; r < 2 * RES; ++r)
#pragma unroll
            for (int d = 0; d < 3; ++d) wres[r][d] = wload(wbase(0) + ((r * 3 + d) * NT) * 1024);
    }
    if (tid < R * 4 * sk::kZeroPos)
        *reinterpret_cast<uint4*>(lds + (tid / (4 * sk::kZeroPos)) * plane_bytes + zero_addr + (tid % (4 * sk::kZeroPos)) * 16) = make_uint4(0, 0, 0, 0);
    if constexpr (RES > 0 && WL > 0) {   // rows RES .. RES + WL - 1 of chunk 0: one copy per workgroup
        for (int i = tid; i < WL * 6 * 64; i += 256)
            *reinterpret_cast<uint4*>(lds + R * plane_bytes + i * 16) =
                *reinterpret_cast<const uint4*>(a.wpk + RES * 6 * 1024 + i * 16);
    }
    issue_dma(0, ch0, false, 0);
    if constexpr (RES == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) a0[d] = wload(wbase(ch0) + (d * NT) * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!SPLIT) activate(0, ch0, false, 0);
    __syncthreads();

    int step = 0, k = 0, ch = ch0, rot = 0;   // k: position of the phase in its step's chunk order
    SK_T_DECL
    // MIX8: the phase kind is a compile-time value (see conv3_kernel): two steps = (fp16, fp8), (fp8, fp16)
    constexpr int kSub = MIX8 ? 4 : 1;
    for (int ph0 = 0; ph0 < nphases; ph0 += kSub)
#pragma unroll
    for (int sub = 0; sub < kSub; ++sub) {
        const int ph = ph0 + sub;
        if (MIX8 && ph >= nphases) break;   // an odd number of steps: the last pair of the four does not exist
        const int x0 = xa + step * XS;
        if (k == 0) {
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int o = 0; o < XS; ++o)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[p][o][i][0] = acc[p][o][i][1] = *reinterpret_cast<const f32x4*>(biasp + 16 * i);
        }
        // ---------------- MFMA over the 27 taps of this chunk ---------------------------
        {
            const unsigned wch = wbase(ch);
            int pslot[R];
#pragma unroll
            for (int i = 0; i < R; ++i) pslot[i] = ((rot + i) % R) * plane_bytes;

            // one (dy,dz) row for cout half `ks` (= i): K = the chunk's 32 channels in ONE instruction
            auto compute = [&](int dydz, int ks, const half8 (&afr)[3]) {
                const int dz = dydz % 3 - 1;
                const int tapoff = (dydz / 3 - 1) * pitch + dz;
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    // the second voxel half sits a.jstep positions on: 16 (the same row: same swizzle term, + 1 KiB) or,
                    // for the 16-wide rectangle patches, one pitch (the next row)
                    const int q = q_row[p] + tapoff;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int qj = q + j * a.jstep;
                        int addr = (qj * 4 + (g ^ (((qj >> 2) & 1) << 1))) * 16;
                        if (dz < 0) addr = zlo(p, j) ? sk::zero_of(zero_addr, addr) : addr;
                        if (dz > 0) addr = zhi(p, j) ? sk::zero_of(zero_addr, addr) : addr;
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            const half8 bfr = *reinterpret_cast<const half8*>(lds + pslot[i] + addr);
#pragma unroll
                            for (int d = 0; d < 3; ++d) {
                                const int o = i - d;  // x_in = x_out + (d - 1)
                                if (o >= 0 && o < XS)
                                    acc[p][o][ks][j] =
                                        SK_MFMA_16x16x32_T16(afr[d], bfr, acc[p][o][ks][j], 0, 0, 0);
                            }
                        }
                    }
                }
            };
            // hipcc schedules the ds_read / MFMA interleave of a (dydz, ks) body itself (pinning
            // it with sched_barrier measured 8 % slower); the weight fragments of the next body
            // are requested one body ahead.
            // Fully unrolling the 9 tap rows lets hipcc hoist the next row's loads: +1.5 % for
            // COUT 32 / 64, -8 % for COUT 128 (code size), measured A/B on one device.
            constexpr int kTapUnroll = (NT <= 1) ? 9 : 1;
            // Without resident weight rows the registers hold the B fragments of TWO tap rows instead: a row's 12
            // fragments are read once for both cout halves, and the next row's are requested before this row's 48
            // MFMAs.  tools/conv_phase_timing.py: a wave spends 70-73 % of its life in the MFMA phase and that phase
            // takes 2.1x its MFMA cycles -- 2x is the share of the pipe when the co-resident wave multiplies too, the
            // rest is a wave that has the SIMD to itself exposing one LDS latency per body.  dec0.0 (two chunks):
            // 1.316 -> 1.228 ms per 8 tiles (-6.7 %); the single-chunk layers keep RES = 2 (resident rows beat it, +3 %).
            const bool f8phase = MIX8 && (sub == 1 || sub == 2);   // (= a.chinfo[ch] & 4)
            if (f8phase) {
                if constexpr (MIX8) {
                    typedef int v8i __attribute__((ext_vector_type(8)));
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    // this lane's K block of an instruction: g >> 1 = which tap row of the pair, g & 1 = x8 (0) | lo8 (1):
                    // 32 contiguous bytes of the staged position (two 16-byte slots; the plane swizzle exchanges the 32-byte halves)
                    const int halfsel = (g & 1) * 2;
                    const int sa = a.w8_scale, sb = 0x70707070;   // E8M0: weights 2^-b (host), activations 2^-15
                    // the six weight fragments of a tap-row pair (both cout halves x three x taps: 48 registers)
                    auto wload8 = [&](int rp, v8i (&af)[2][3]) {
                        const unsigned wrp = (unsigned)a.w8_off + (unsigned)(rp * 6) * 2048u + wlane * 2;
#pragma unroll
                        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                            for (int d = 0; d < 3; ++d) {
                                const unsigned wo = wrp + (unsigned)((i2 * 3 + d) * 2048);
                                const u32x4 w0 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wo, 0, 0);
                                const u32x4 w1 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wo + 16, 0, 0);
                                af[i2][d] = v8i{(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
                            }
                    };
                    auto mma8 = [&](int rp, const v8i (&af)[2][3]) {
                        const int rr = min(2 * rp + (g >> 1), 8);   // (the tenth row does not exist: its weights are zero)
                        const int dz = rr % 3 - 1;
                        const int tapoff = (rr / 3 - 1) * pitch + dz;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int qj = q_row[0] + tapoff + j * a.jstep;
                            int addr = (qj * 4 + (halfsel ^ (((qj >> 2) & 1) << 1))) * 16;
                            if (dz < 0 && zlo(0, j)) addr = sk::zero_of(zero_addr, addr);
                            if (dz > 0 && zhi(0, j)) addr = sk::zero_of(zero_addr, addr);
#pragma unroll
                            for (int i = 0; i < R; ++i) {
                                const u32x4 lo4 = *reinterpret_cast<const u32x4*>(lds + pslot[i] + addr);
                                const u32x4 hi4 = *reinterpret_cast<const u32x4*>(lds + pslot[i] + addr + 16);
                                const v8i bf = v8i{(int)lo4[0], (int)lo4[1], (int)lo4[2], (int)lo4[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
#pragma unroll
                                for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                                    for (int d = 0; d < 3; ++d) {
                                        const int o = i - d;
                                        if (o >= 0 && o < XS)
                                            acc[0][o][i2][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i2][d], bf, acc[0][o][i2][j], 0, 0,
                                                                                                               0, sa, 0, sb);
                                    }
                            }
                        }
                    };
                    // (one body per pair, not unrolled: hipcc hoists the next pair's loads when it is, at 42 spilled registers for
                    // XS = 4, and the launch takes the same time -- profiles/r04_ab_mix8_variants.txt; nor does XS = 3 pay: +4.5 %)
#pragma unroll 1
                    for (int rp = 0; rp < 5; ++rp) {
                        v8i af[2][3];
                        wload8(rp, af);
                        mma8(rp, af);
                    }
                }
            } else if constexpr (RES == 0 && NT == 1) {
                half8 bb[2][2][R];
                auto load_row = [&](int dydz, half8 (&dst)[2][R]) {
                    const int dz = dydz % 3 - 1;
                    const int q = q_row[0] + (dydz / 3 - 1) * pitch + dz;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int qj = q + j * a.jstep;
                        int addr = (qj * 4 + (g ^ (((qj >> 2) & 1) << 1))) * 16;
                        if (dz < 0) addr = zlo(0, j) ? sk::zero_of(zero_addr, addr) : addr;
                        if (dz > 0) addr = zhi(0, j) ? sk::zero_of(zero_addr, addr) : addr;
#pragma unroll
                        for (int i = 0; i < R; ++i) dst[j][i] = *reinterpret_cast<const half8*>(lds + pslot[i] + addr);
                    }
                };
                auto mma_row = [&](int ks, const half8 (&afr)[3], const half8 (&src)[2][R]) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < R; ++i)
#pragma unroll
                            for (int d = 0; d < 3; ++d) {
                                const int o = i - d;
                                if (o >= 0 && o < XS)
                                    acc[0][o][ks][j] = SK_MFMA_16x16x32_T16(afr[d], src[j][i], acc[0][o][ks][j], 0, 0, 0);
                            }
                };
                load_row(0, bb[0]);
#pragma unroll
                for (int dydz = 0; dydz < 9; ++dydz) {
                    const unsigned wrow = wch + (unsigned)(dydz * 2) * (3 * NT) * 1024;
#pragma unroll
                    for (int d = 0; d < 3; ++d) a1[d] = wload(wrow + ((3 + d) * NT) * 1024);
                    if (dydz < 8) load_row(dydz + 1, bb[(dydz + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_row(0, a0, bb[dydz & 1]);
                    if (dydz < 8) {
#pragma unroll
                        for (int d = 0; d < 3; ++d) a0[d] = wload(wrow + ((6 + d) * NT) * 1024);
                    }
                    mma_row(1, a1, bb[dydz & 1]);
                }
            } else
#pragma unroll kTapUnroll
            for (int dydz = 0; dydz < 9; ++dydz) {
                const unsigned wrow = wch + (unsigned)((SK_ABL(a, 2) ? 0 : dydz) * 2) * (3 * NT) * 1024;
                if constexpr (RES > 0) {
                    if (dydz < RES) {
                        compute(dydz, 0, wres[2 * dydz]);
                        if (dydz == RES - 1 && WL == 0) {  // first streamed row's ks = 0 fragments, one body ahead
#pragma unroll
                            for (int d = 0; d < 3; ++d)
                                a0[d] = wload(wrow + ((6 + d) * NT) * 1024);
                        }
                        compute(dydz, 1, wres[2 * dydz + 1]);
                        continue;
                    }
                    if (dydz < RES + WL) {   // rows whose fragments sit in LDS behind the ring
                        const char* wl = lds + R * plane_bytes + (dydz - RES) * 6144 + lane * 16;
                        half8 l0[3], l1[3];
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            l0[d] = *reinterpret_cast<const half8*>(wl + d * 1024);
                            l1[d] = *reinterpret_cast<const half8*>(wl + (3 + d) * 1024);
                        }
                        compute(dydz, 0, l0);
                        if (dydz == RES + WL - 1) {
#pragma unroll
                            for (int d = 0; d < 3; ++d)
                                a0[d] = wload(wrow + ((6 + d) * NT) * 1024);
                        }
                        compute(dydz, 1, l1);
                        continue;
                    }
                }
                if (!SK_ABL(a, 32) || dydz == 0) {
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                        a1[d] = wload(wrow + ((3 + d) * NT) * 1024);
                }
                compute(dydz, 0, a0);
                if (dydz < 8 && !SK_ABL(a, 32)) {
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                        a0[d] = wload(wrow + ((6 + d) * NT) * 1024);
                }
                compute(dydz, 1, a1);
            }
        }

        // ---------------- hand the LDS planes to the next phase ---------------------------
        const bool step_done = (k == nck - 1);
        int nstep = step, nk = k + 1;
        if (step_done) {
            nstep = step + 1;
            nk = 0;
        }
        const int nch = chunk_of(nstep, nk);
        const bool have_next = ph + 1 < nphases;
        // same staged data (source and byte offset in the voxel line), next step: its planes XS, XS + 1 stay
        const bool reuse_n = step_done && (SPLIT ? ((a.chinfo[nch] ^ a.chinfo[ch]) & ~6u) == 0 : nch == ch);
        const int rot_n = reuse_n ? (rot + XS) % R : rot;
        SK_T(0)           // MFMA phase
        __syncthreads();  // every wave is done reading the planes about to be overwritten
        SK_T(1)           // barrier: waiting for the slowest wave of the workgroup
        if (have_next) {
            if (!(a.chinfo[nch] & 2)) issue_dma(nstep, nch, reuse_n, rot_n);   // bit 1: the next chunk multiplies the SAME staged planes
            if constexpr (RES == 0) {
#pragma unroll
                for (int d = 0; d < 3; ++d) a0[d] = wload(wbase(nch) + (d * NT) * 1024);
            }
            // The DMA must have landed before this wave passes the closing barrier.  vmcnt retires in issue
            // order, so after an epilogue that issues a FIXED number of stores (invalid lanes / planes store to
            // a scratch line instead of branching) `vmcnt(that number)` means "everything older than the
            // stores -- the DMA and the weight fragments -- has landed": the DMA latency hides behind the
            // epilogue's transposes and the stores still drain under the next phase's MFMAs.
            // Measured per layer (same device, A/B): -4 % time for COUT 32, +4 % for COUT 64 (two column tiles per
            // wave: a longer epilogue), neutral for COUT 128 -> late wait for NT != 2 only.
            if (!kLateWait || !step_done || SK_ABL(a, ~0)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }

        SK_T(2)           // LDS-DMA issue + weight prefetch (+ the landing wait where it is not deferred)
        // ---------------- epilogue (overlaps the DMA): raw fp16 store + GroupNorm partials ------
        // The accumulator tile is [cout rows in registers][voxel columns on lanes]; the output
        // is channels-last.  Each wave transposes its 32x32 tile through a private 2 KiB LDS
        // pad (16-B chunks XOR-swizzled) so that every lane stores 16 contiguous bytes and one
        // store instruction writes 1 KiB of whole 64-B voxel lines (4 lanes per line).
        if (step_done) {
            char* outw = outb + (long long)x0 * out_plane;   // the step's XS output planes: one descriptor
            const __amdgpu_buffer_rsrc_t rout = sk::make_rsrc(outw, (unsigned)(XS * out_plane));
#pragma unroll
            for (int p = 0; p < P; ++p) {
#pragma unroll
                for (int o = 0; o < XS; ++o) {
                    const int x = x0 + o;
#pragma unroll
                    for (int part = 0; part < (SPLIT ? 2 : 1); ++part) {   // SPLIT: the hi halves, then the lo halves
                        // The accumulator layout gives a lane 4 channels (16 i + 4 g ..) of voxel c16 (j = 0) and of voxel
                        // 16 + c16 (j = 1); the lane 16 further on holds the next 4 channels of the same two voxels.
                        // v_permlane16_swap_b32 (gfx950) exchanges the odd 16-lane rows of one register with the even rows
                        // of another: afterwards even rows hold 8 consecutive channels of voxel c16, odd rows 8 channels of
                        // voxel 16 + c16 -- 16 contiguous bytes per lane with no trip through LDS (the transposing pad
                        // cost two LDS round trips per plane: 17 % of a wave's cycles in tools/conv_phase_timing.py).
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            unsigned d[2][2];   // [j][dword]: channels 16 i + 4 g + (0,1 | 2,3)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const f32x4 r = acc[p][o][i][j];
                                half4 hv = {(t16)r[0], (t16)r[1], (t16)r[2], (t16)r[3]};
                                if (part == 1)   // lo = fp16(v - hi): exact difference, rounded once
                                    hv = half4{(t16)(r[0] - (float)hv[0]), (t16)(r[1] - (float)hv[1]),
                                               (t16)(r[2] - (float)hv[2]), (t16)(r[3] - (float)hv[3])};
                                const uint2 u = __builtin_bit_cast(uint2, hv);
                                d[j][0] = u.x;
                                d[j][1] = u.y;
                                if (SPLIT) {
                                    if (part == 0 && vvalid(p, j) && x < xb) {   // split: statistics of the fp32 accumulators
                                        gsum[i] += (r[0] + r[1]) + (r[2] + r[3]);
                                        gsq[i] += (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
                                    }
                                } else {
                                    // statistics of the values as STORED (16-bit), on v_dot2c: exact products, fp32 sums,
                                    // 4 instructions per 4 values instead of 4 multiplies + 7 adds; a voxel outside the
                                    // tile contributes zeros
                                    const bool in = vvalid(p, j) && x < xb;
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
                            // this lane now owns channels 16 i + 8 (g >> 1) .. +7 of voxel c16 + 16 (g & 1)
                            const uint4 line = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                            if (!SK_ABL(a, 4)) {
                                const bool xbox = !a.has_box || (x >= a.box_lo[0] && x < a.box_hi[0]);   // wave-uniform
                                const bool sok = x < xb && xbox && ((sboxm >> p) & 1u) && svox[p] >= 0;
                                char* dst = outb + (long long)x * out_plane + (long long)svox[p] * kOvs + wn * 64 + part * (COUT * 2) +
                                            32 * i + 16 * (g >> 1);
                                // always issued (the counted wait relies on it); a masked lane's offset is out of range: dropped
                                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                                const u32x4 lv = {line.x, line.y, line.z, line.w};
                                __builtin_amdgcn_raw_buffer_store_b128(lv, rout, sok ? (unsigned)(dst - outw) : sk::kOob, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        SK_T(3)           // epilogue
        if (have_next) {
            if (kLateWait && step_done && !SK_ABL(a, ~0)) {
                constexpr int kStores = P * XS * 2 * (SPLIT ? 2 : 1);  // global stores the epilogue just issued
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kStores) : "memory");
            }
            SK_T(4)       // deferred landing wait
            if (!SPLIT && !(a.chinfo[nch] & 2)) activate(nstep, nch, reuse_n, rot_n);   // own DMA has landed (waits above)
            SK_T(5)       // in-LDS activation
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            SK_T(6)       // closing barrier
        }
        step = nstep;
        k = nk;
        ch = nch;
        rot = rot_n;
    }

    SK_T_DUMP(a, w, lane)
    // ---- block-level reduction of the GroupNorm partials ------------------------------------
    if (a.partial) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);  // [4 waves][8 quads of the wave's cout tile][2]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float s = gsum[i], ss = gsq[i];
#pragma unroll
            for (int m = 8; m > 0; m >>= 1) {   // the 16 lanes that share g
                s += __shfl_xor(s, m);
                ss += __shfl_xor(ss, m);
            }
            if (c16 == 0) {
                red[(w * 8 + 4 * i + g) * 2 + 0] = s;
                red[(w * 8 + 4 * i + g) * 2 + 1] = ss;
            }
        }
        __syncthreads();
        if (tid < NT * 16) {
            // channel quad Q = cout/4 = 8*nt + k ; waves with wn == nt: w = wm*NT + nt
            const int nt = tid / 16, k2 = tid % 16;
            float t = 0.0f;
#pragma unroll
            for (int g = 0; g < 4 / NT; ++g) t += red[(g * NT + nt) * 16 + k2];
            a.partial[((long long)b * nblk + block_in_batch) * (NT * 16) + tid] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------
// conv3_px_kernel (round 3): the single-chunk COUT-32 layers (32 -> 32: enc0.1, dec0.1) marching along x by INPUT
// plane -- different wave work per step, not different knobs.
//
// conv3_m16_kernel's step is serial inside a workgroup: MFMAs over a six-plane ring, barrier, the NEXT step's LDS-DMA
// (11 pieces per wave), 24 weight-fragment loads streamed behind the MFMAs, epilogue (8 stores), landing wait, barrier --
// 43 vector-memory instructions per wave and step, and DESIGN.md section 8 prices each at 100-350 cycles of wave time
// (profiles/r03_conv_sq_counters.json: MFMA busy 0.555 against 0.70-0.75 for the COUT 64 / 128 kernels).  Here
//   * a step consumes TWO input planes; input plane x contributes its 27 taps to the output planes x-1, x, x+1, so four
//     rolling accumulator planes are live (64 registers, as before) and two output planes complete per step;
//   * what a step reads from the ring is only its own two planes: the ring holds 4 plane slots (2 being read, 2 landing)
//     instead of 6, and the LDS-DMA of the planes of step s+1 is issued at the START of step s -- a whole step of MFMAs
//     (~3.5 k cycles) to land in, nothing serial about it;
//   * the LDS the ring gives back holds the weights: the first half tap rows in registers, the others in LDS (one copy
//     per workgroup, loaded once per x-chunk) -- NO weight fragment is loaded in the loop: a wave issues 5-6 LDS-DMA
//     pieces + 4 stores per 216 MFMAs where conv3_m16_kernel issues 43 vector-memory instructions per 432;
//   * B fragments: one ds_read_b128 feeds the three x-taps of both cout halves (6 MFMAs), 0.17 reads per MFMA against
//     0.25; the LDS-resident weight rows add 0.14.
// The summation order of an output voxel (x-tap major) is a function of the voxel alone: batch- and chunk-invariant
// like conv3_m16_kernel's, not bit-identical to it (asserted against torch to one fp16 ulp like every conv kernel,
// bit-exact on integer operands).
// Store box (Conv3Args.box, sk_conv3d_box): only the voxels inside it are stored -- the statistics still cover the
// whole tile.  The last block's conv (dec0.1) is read by the heads on 28 % of the tile (the scatter's box).
template <int RESH, int NPOSP>
__global__ void __launch_bounds__(256, 2) conv3_px_kernel(Conv3Args a) {
    constexpr int NSLOT = 4;
    constexpr int plane_bytes = NPOSP * kPosBytes;   // compile-time: a slot's offset is an immediate of its ds_read_b128
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // column tile of this wave (32 voxels of the 128-voxel patch)
    const int c16 = lane & 15, g = lane >> 4;

    int blk = blockIdx.x;   // XCD-aware order, see conv3_kernel
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

    // ---- patch geometry (conv3_m16_kernel's, one column tile per wave) ---------------------------
    const int pitch = a.pitch;
    int off, ybase, zbase, q_row, out_vox0, tile_nvox;
    int svy, svz;             // (y, z) of the voxel this lane STORES: column c16 + 16 (g & 1) of the wave's tile
    unsigned vflags = 0;      // bit j: voxel 16 j + c16 on the z = 0 face | << 8: on the z = Zt-1 face | << 16: inside the tile
    auto zlo = [&](int j) { return (vflags >> j) & 1u; };
    auto zhi = [&](int j) { return (vflags >> (8 + j)) & 1u; };
    auto vvalid = [&](int j) { return (vflags >> (16 + j)) & 1u; };
    // linear mode only (Zt <= 40, conv3_m16_kernel's comment): region position q <-> in-plane voxel v0 - Zt - 1 + q
    const int needed = kPatch + 2 * a.Zt + 2;   // positions a plane really holds; NPOSP rounds it up to a DMA granule
    {
        const int v0 = patch * kPatch;
        off = v0 - a.Zt - 1;
        ybase = 0;
        zbase = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int v = v0 + 32 * w + 16 * j + c16;
            const int vy = v / a.Zt, vz = v - vy * a.Zt;
            vflags |= (unsigned)(v < a.Yt * a.Zt) << (16 + j);
            vflags |= (unsigned)(vz == 0) << j;
            vflags |= (unsigned)(vz == a.Zt - 1) << (8 + j);
            if (j == 0) q_row = v - off;
        }
        out_vox0 = v0 + 32 * w;
        tile_nvox = a.Yt * a.Zt;
        const int sv = out_vox0 + c16 + 16 * (g & 1);
        svy = sv / a.Zt;
        svz = sv - svy * a.Zt;
    }
    const bool sbox = !a.has_box || (svy >= a.box_lo[1] && svy < a.box_hi[1] && svz >= a.box_lo[2] && svz < a.box_hi[2]);

    // ---- LDS-DMA bookkeeping: this lane's slots of a plane (conv3_m16_kernel's swizzle) -----------
    constexpr int ndma = NPOSP / 16;
    int d_vox[kMaxDma];
    const int d_cs = ((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16;
#pragma unroll
    for (int k = 0; k < kMaxDma; ++k) {
        const int t = w + 4 * k;
        const int q = (64 * t + lane) >> 2;
        const int Pq = q + off;
        const int y = ybase + (Pq >= 0 ? Pq / pitch : -1), z = zbase + (Pq >= 0 ? Pq % pitch : 0);
        // positions >= needed are padding that the LDS-DMA never writes (d_vox -2: the lane sits out of the instruction):
        // two of them hold the bias / the GroupNorm coefficients of a raw source, the last four are the zero window
        const bool ok = (t < ndma) && y >= 0 && y < a.Yt && z >= 0 && z < a.Zt;
        d_vox[k] = q >= needed ? -2 : (ok ? (y * a.Zt + z) * 64 + d_cs : -1);   // the BYTE offset of this lane's 16-byte piece in a plane
    }

    const int xa = xc * a.XC;
    const int xb = min(xa + a.XC, a.Xt);
    const int n = xb - xa;                 // output planes xa .. xb-1 of this workgroup
    const int nin = n + 2;                 // input planes t = 0 .. nin-1 <-> x = xa - 1 + t; plane t lives in slot t & 3
    // The last four positions of every slot are a 256-byte window of zeros (never written by the LDS-DMA).  A tap that
    // leaves the tile through a z face reads zeros at `zero_addr + (its own address & 255)`: the SAME banks its data
    // read would have used.  With one shared zero line instead, the four lanes of a z-face voxel (one per K group, one
    // in each ds_read_b128 lane group) each collide with another lane's banks: 8 LDS cycles instead of 4 for the reads
    // of six of the nine tap rows, SQ_LDS_BANK_CONFLICT 18-35 % of the LDS-active cycles in every conv3 kernel
    // (simulated on the documented lane groups: 6.13 cycles per B read on average against 4.00).
    constexpr int zero_addr = (NPOSP - 4) * kPosBytes;
    static_assert(zero_addr % 256 == 0, "the zero window must cover the 64 banks once");
    constexpr int kOvs = 64;               // bytes per output voxel (32 channels)
    const long long out_plane = (long long)a.Yt * a.Zt * kOvs;
    char* outb = a.out + (long long)b * a.Xt * out_plane;
    const SrcDev s0 = a.src[0];
    const char* srcb = s0.data + (long long)b * s0.batch;

    auto issue_plane = [&](int t) {
        const int x = xa - 1 + t;
        if (x < 0 || x >= a.Xt) return;   // outside the tile: the steps that would read it skip their MFMAs
        if (SK_ABL(a, 1) && t >= NSLOT) return;   // timing experiment (-DSK_TUNING): no LDS-DMA after the first planes
        const __amdgpu_buffer_rsrc_t rsrc = sk::make_rsrc(srcb + (long long)x * s0.plane, (unsigned)s0.plane);
        char* lbase = lds + (t & (NSLOT - 1)) * plane_bytes;
#pragma unroll
        for (int k = 0; k < kMaxDma; ++k) {
            const int tt = w + 4 * k;
            if (tt < ndma && d_vox[k] != -2) {   // padding lanes are masked out of the plane's last piece (EXEC)
                const unsigned voff = d_vox[k] >= 0 ? (unsigned)d_vox[k] : sk::kOob;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lbase + tt * 1024), 16, voff, 0, 0, 0);
            }
        }
    };
    // GroupNorm affine + SiLU of a RAW source in LDS, by the lane that staged the piece (conv3_m16_kernel's `activate`)
    const float* af = a.act[0];
    auto activate_plane = [&](int t) {
        const int x = xa - 1 + t;
        if (x < 0 || x >= a.Xt) return;
        // the coefficients of this lane's 8 channels, from the copy in LDS (a global load here cost the wave a memory
        // latency and a vmcnt(0) per plane; keeping them in registers costs 16 of the 256)
        const float* cfa = reinterpret_cast<const float*>(lds + needed * kPosBytes) + (d_cs / 16) * 8;
        const float* cfb = reinterpret_cast<const float*>(lds + 2 * plane_bytes + needed * kPosBytes) + (d_cs / 16) * 8;
        float ga[8], gb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ga[j] = cfa[j];
            gb[j] = cfb[j];
        }
        char* lbase = lds + (t & (NSLOT - 1)) * plane_bytes;
        half8 v[kMaxDma];
#pragma unroll
        for (int k = 0; k < kMaxDma; ++k)
            if (w + 4 * k < ndma) v[k] = *reinterpret_cast<const half8*>(lbase + (w + 4 * k) * 1024 + lane * 16);
#pragma unroll
        for (int k = 0; k < kMaxDma; ++k) {
            const int tt = w + 4 * k;
            if (tt < ndma) {
                const bool real = d_vox[k] >= 0;
                half8 r;
#pragma unroll
                for (int e = 0; e < 8; ++e) {   // gn_silu_kernel's arithmetic, op for op
                    const float y = fmaf(ga[e], (float)v[k][e], gb[e]);
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-y));
                    const t16 act = sk::round_t16(y * sg);   // unconditional: the asm inside must not sit behind a per-element branch
                    r[e] = real ? act : v[k][e];
                }
                if (d_vox[k] != -2) *reinterpret_cast<half8*>(lbase + tt * 1024 + lane * 16) = r;   // not the padding
            }
        }
    };

    // ---- weights: half rows (tap row dydz, cout half i) 0 .. RESH-1 in registers, RESH .. 17 in LDS behind the ring ----
    // fragment of (row dydz, cout half i, x tap d): ((dydz * 2 + i) * 3 + d) KiB into the packed weight
    const __amdgpu_buffer_rsrc_t wrsrc = sk::make_rsrc(a.wpk, 54u * 1024u);
    half8 wres[RESH][3];
#pragma unroll
    for (int r = 0; r < RESH; ++r)
#pragma unroll
        for (int d = 0; d < 3; ++d)
            wres[r][d] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16, (r * 3 + d) * 1024, 0));
    char* wlds = lds + NSLOT * plane_bytes;
    for (int i = tid; i < (18 - RESH) * 3 * 64; i += 256)
        *reinterpret_cast<uint4*>(wlds + i * 16) = *reinterpret_cast<const uint4*>(a.wpk + RESH * 3 * 1024 + i * 16);
    // padding positions of a slot: needed, needed + 1 (128 bytes: slot 0 the GroupNorm scales of a raw source, slot 1 the
    // bias, slot 2 the GroupNorm shifts) | the zero window NPOSP - 4 .. NPOSP - 1
    if (tid < NSLOT * 16)
        *reinterpret_cast<uint4*>(lds + (tid >> 4) * plane_bytes + zero_addr + (tid & 15) * 16) = make_uint4(0, 0, 0, 0);
    if (tid >= 128 && tid < 160) reinterpret_cast<float*>(lds + plane_bytes + needed * kPosBytes)[tid - 128] = a.bias[tid - 128];
    if (af && tid >= 64 && tid < 128) {   // (2, 32) coefficients of this batch item
        const int c = tid - 64;
        reinterpret_cast<float*>(lds + (c < 32 ? 0 : 2 * plane_bytes) + needed * kPosBytes)[c & 31] = af[(long long)b * 64 + c];
    }

    SK_T_DECL
    int issued = 0;
    auto issue_upto = [&](int lim) {   // planes [issued, min(nin, lim))
        const int hi = min(nin, lim);
        const int lo = issued;
        for (int t = lo; t < hi; ++t) issue_plane(t);
        issued = max(issued, hi);
        return lo;
    };
    issue_upto(NSLOT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (af) {
        __syncthreads();   // the coefficients in LDS were written by other waves
        for (int t = 0; t < issued; ++t) activate_plane(t);
    }
    __syncthreads();

    // ---- accumulators ----------------------------------------------------------------------------------------
    // element r of [i][j]: cout 16 i + 4 g + r, voxel 16 j + c16 of the wave's column tile
    f32x4 P0[2][2], P1[2][2], Q0[2][2], Q1[2][2];
    // the bias (the accumulators' initial value) is re-read from its copy in LDS -- padding positions of slot 1 -- at every
    // reset: eight registers less in the loop
    const float* lbias = reinterpret_cast<const float*>(lds + plane_bytes + needed * kPosBytes) + 4 * g;
    auto reset = [&](f32x4 (&o)[2][2]) {
        o[0][0] = o[0][1] = *reinterpret_cast<const f32x4*>(lbias);
        o[1][0] = o[1][1] = *reinterpret_cast<const f32x4*>(lbias + 16);
    };
    reset(P0);   // (behind the barrier above: the bias in LDS was written by another wave)
    reset(P1);
    reset(Q0);
    reset(Q1);
    float gsum[2] = {0.0f, 0.0f}, gsq[2] = {0.0f, 0.0f};

    auto baddr = [&](int dydz, int j) -> int {
        const int dz = dydz % 3 - 1;
        const int q = q_row + (dydz / 3 - 1) * pitch + dz;
        int addr = (q * 4 + (g ^ (((q >> 2) & 1) << 1))) * 16 + 1024 * j;
        if (dz < 0) addr = zlo(j) ? zero_addr + (addr & 255) : addr;
        if (dz > 0) addr = zhi(j) ? zero_addr + (addr & 255) : addr;
        return addr;
    };
    auto wfrag = [&](int dydz, int i, half8 (&dst)[3]) {
        if (2 * dydz + i < RESH) {
#pragma unroll
            for (int d = 0; d < 3; ++d) dst[d] = wres[2 * dydz + i][d];
        } else {
            const char* p = wlds + (2 * dydz + i - RESH) * 3 * 1024 + lane * 16;
#pragma unroll
            for (int d = 0; d < 3; ++d) dst[d] = *reinterpret_cast<const half8*>(p + d * 1024);
        }
    };

    // A step over the input planes A (slot sA) and B = A + 1 (slot sB).  oA1 / oA / oB / oB1: the accumulators of the
    // output planes A-1, A, B, B+1.  Tap d of a weight row multiplies x_in = x_out + d - 1.
    auto pair_step = [&](auto SA, auto SB, f32x4 (&oA1)[2][2], f32x4 (&oA)[2][2], f32x4 (&oB)[2][2], f32x4 (&oB1)[2][2]) {
        // compile-time slots: the 18 tap addresses of the patch (plane-relative, loop-invariant) serve both planes of
        // every step through the immediate offset of ds_read_b128
        const char* pa = lds + decltype(SA)::value * plane_bytes;
        const char* pb = lds + decltype(SB)::value * plane_bytes;
        half8 bq[2][2][2];   // [buffer][plane][j]: the B fragments of a tap row, one row ahead
        half8 wq[2][3];      // [buffer][d]: the weight fragments of a half row (cout half i of a tap row), one half row ahead
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ad = baddr(0, j);
            bq[0][0][j] = *reinterpret_cast<const half8*>(pa + ad);
            bq[0][1][j] = *reinterpret_cast<const half8*>(pb + ad);
        }
        wfrag(0, 0, wq[0]);
        // The 18 half rows of a step, 12 MFMAs each (192 cycles of the matrix pipe).  The LDS reads of half row h + 1 -- its
        // three weight fragments when its tap row lives in LDS, and the four B fragments of the next tap row -- are issued
        // in the FIRST MFMA gaps of half row h, in the order half row h + 1 consumes them, so the youngest read is 80+
        // cycles old (and not needed before the seventh MFMA) when half row h + 1 starts.  The order is pinned: left to
        // itself the scheduler spread the reads to the END of the half row and every half row began with an
        // `s_waitcnt lgkmcnt(0)` on a read issued one MFMA earlier (SQ_WAIT_ANY 38 % of the wave-cycles).
#pragma unroll
        for (int dydz = 0; dydz < 9; ++dydz) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int h = dydz * 2 + i;   // its weights sit in wq[h & 1], its B fragments in bq[dydz & 1]
                const half8(&W)[3] = wq[h & 1];
                const int ndy = i == 0 ? dydz : dydz + 1, ni = i ^ 1;         // the next half row
                bool wread = ndy < 9 && 2 * ndy + ni >= RESH;                  // ... reads its weights from LDS
                bool bread = i == 1 && dydz < 8;                               // ... starts a new tap row: B fragments
                if (SK_PX_ABL(16) && wread) {   // timing experiment: no LDS weight reads (a resident half row instead)
                    wread = false;
                    wfrag((2 * ndy + ni) % RESH / 2, (2 * ndy + ni) % RESH % 2, wq[(h + 1) & 1]);
                } else if (ndy < 9 && !wread) {
                    wfrag(ndy, ni, wq[(h + 1) & 1]);                           // resident row: register names only
                }
                if (SK_PX_ABL(32) && bread) {   // timing experiment: no B fragment reads after the first tap row
                    bread = false;
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) bq[(dydz + 1) & 1][pl][jj] = bq[dydz & 1][pl][jj];
                }
                // read k of the next half row, in consumption order: W0, A0, B0, W1, W2, A1, B1
                auto next_read = [&](int k) {
                    const char* wp = wlds + (2 * ndy + ni - RESH) * 3 * 1024 + lane * 16;
                    int kk = k;
                    if (!wread) kk = (k == 0 ? 1 : k == 1 ? 2 : k == 2 ? 5 : 6);   // B fragments only: A0, B0, A1, B1
                    if (!bread && kk > 0) kk = (kk == 1 ? 3 : 4);                  // weights only: W0, W1, W2
                    switch (kk) {
                        case 0: wq[(h + 1) & 1][0] = *reinterpret_cast<const half8*>(wp); break;
                        case 1: bq[(dydz + 1) & 1][0][0] = *reinterpret_cast<const half8*>(pa + baddr(dydz + 1, 0)); break;
                        case 2: bq[(dydz + 1) & 1][1][0] = *reinterpret_cast<const half8*>(pb + baddr(dydz + 1, 0)); break;
                        case 3: wq[(h + 1) & 1][1] = *reinterpret_cast<const half8*>(wp + 1024); break;
                        case 4: wq[(h + 1) & 1][2] = *reinterpret_cast<const half8*>(wp + 2048); break;
                        case 5: bq[(dydz + 1) & 1][0][1] = *reinterpret_cast<const half8*>(pa + baddr(dydz + 1, 1)); break;
                        default: bq[(dydz + 1) & 1][1][1] = *reinterpret_cast<const half8*>(pb + baddr(dydz + 1, 1)); break;
                    }
                };
                const int nreads = (wread ? 3 : 0) + (bread ? 4 : 0);
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    const int j = m / 6;
                    const half8 fa = bq[dydz & 1][0][j], fb = bq[dydz & 1][1][j];
                    switch (m % 6) {   // tap d of a weight row multiplies x_in = x_out + d - 1
                        case 0: oB[i][j] = SK_MFMA_16x16x32_T16(W[0], fa, oB[i][j], 0, 0, 0); break;
                        case 1: oB1[i][j] = SK_MFMA_16x16x32_T16(W[0], fb, oB1[i][j], 0, 0, 0); break;
                        case 2: oA[i][j] = SK_MFMA_16x16x32_T16(W[1], fa, oA[i][j], 0, 0, 0); break;
                        case 3: oA1[i][j] = SK_MFMA_16x16x32_T16(W[2], fa, oA1[i][j], 0, 0, 0); break;
                        case 4: oB[i][j] = SK_MFMA_16x16x32_T16(W[1], fb, oB[i][j], 0, 0, 0); break;
                        default: oA[i][j] = SK_MFMA_16x16x32_T16(W[2], fb, oA[i][j], 0, 0, 0); break;
                    }
                    if (m < nreads) next_read(m);
                    if (m <= nreads) __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // one input plane, a subset of its three x taps (the planes at the ends of an x-chunk)
    auto single_step = [&](auto D0, auto D1, auto D2, int s, f32x4 (&o0)[2][2], f32x4 (&o1)[2][2], f32x4 (&o2)[2][2]) {
        const char* pa = lds + s * plane_bytes;
#pragma unroll
        for (int dydz = 0; dydz < 9; ++dydz) {
            half8 fa[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) fa[j] = *reinterpret_cast<const half8*>(pa + baddr(dydz, j));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                half8 W[3];
                wfrag(dydz, i, W);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (decltype(D0)::value) o0[i][j] = SK_MFMA_16x16x32_T16(W[0], fa[j], o0[i][j], 0, 0, 0);
                    if constexpr (decltype(D1)::value) o1[i][j] = SK_MFMA_16x16x32_T16(W[1], fa[j], o1[i][j], 0, 0, 0);
                    if constexpr (decltype(D2)::value) o2[i][j] = SK_MFMA_16x16x32_T16(W[2], fa[j], o2[i][j], 0, 0, 0);
                }
            }
        }
    };

    // raw fp16 store of one finished output plane + its GroupNorm partial sums (conv3_m16_kernel's epilogue); returns the
    // number of store instructions issued (0 or 2: the counted landing wait needs it)
    auto finish_plane = [&](f32x4 (&o)[2][2], int x) -> int {
        if (SK_ABL(a, 4)) {   // timing experiment (-DSK_TUNING): no epilogue (the accumulators stay observable through gsum)
            gsum[0] += o[0][0][0] + o[0][1][0];
            gsum[1] += o[1][0][0] + o[1][1][0];
            return 0;
        }
        const bool xbox = !a.has_box || (x >= a.box_lo[0] && x < a.box_hi[0]);   // wave-uniform
        const __amdgpu_buffer_rsrc_t rout = sk::make_rsrc(outb + (long long)x * out_plane, (unsigned)out_plane);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned d[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 r = o[i][j];
                const half4 hv = {(t16)r[0], (t16)r[1], (t16)r[2], (t16)r[3]};
                const uint2 u = __builtin_bit_cast(uint2, hv);
                d[j][0] = u.x;
                d[j][1] = u.y;
                const bool in = vvalid(j);
                const t16x2 z2 = {(t16)0.0f, (t16)0.0f}, one2 = {(t16)1.0f, (t16)1.0f};
                const t16x2 lo2 = in ? t16x2{hv[0], hv[1]} : z2, hi2 = in ? t16x2{hv[2], hv[3]} : z2;
                gsum[i] = SK_DOT2_T16(lo2, one2, gsum[i]);
                gsum[i] = SK_DOT2_T16(hi2, one2, gsum[i]);
                gsq[i] = SK_DOT2_T16(lo2, lo2, gsq[i]);
                gsq[i] = SK_DOT2_T16(hi2, hi2, gsq[i]);
            }
            if (xbox) {
                const auto s0_ = __builtin_amdgcn_permlane16_swap(d[0][0], d[1][0], false, false);
                const auto s1_ = __builtin_amdgcn_permlane16_swap(d[0][1], d[1][1], false, false);
                // this lane now owns channels 16 i + 8 (g >> 1) .. +7 of voxel c16 + 16 (g & 1)
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 lv = {s0_[0], s1_[0], s0_[1], s1_[1]};
                const int vv = c16 + 16 * (g & 1);
                const bool sok = sbox && out_vox0 + vv < tile_nvox;
                const unsigned so = (unsigned)((out_vox0 + vv) * kOvs + 32 * i + 16 * (g >> 1));
                __builtin_amdgcn_raw_buffer_store_b128(lv, rout, sok ? so : sk::kOob, 0, 0);
            }
        }
        return xbox ? 2 : 0;
    };
    // end of a step: this wave's LDS-DMA of the step (issued before its MFMAs, older than its `ns` stores) has landed ->
    // activate what it staged -> barrier: every wave is done with the step's planes and sees the landed ones
    auto end_step = [&](int ns, int t_lo, int t_hi) {
        SK_T(3)
        if (ns == 4)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (ns == 2)
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SK_T(4)
        if (af && !SK_ABL(a, 8)) {
            for (int t = t_lo; t < t_hi; ++t) activate_plane(t);
        }
        SK_T(5)
        if (SK_PX_ABL(64))   // timing experiment: no barrier between the steps
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        SK_T(1)
    };
    typedef std::integral_constant<bool, true> T_;
    typedef std::integral_constant<bool, false> F_;

    // ---- the march -----------------------------------------------------------------------------------------------
    // P0 / P1: the two oldest live output planes (P0 = plane xa - 1 at first: never stored), Q0 / Q1 the next two
    // (-DSK_TIMING marks: 0 MFMA | 1 closing barrier | 2 LDS-DMA issue | 3 epilogue | 4 landing wait | 5 in-LDS activation |
    //  6 workgroup prologue | 7 the single-plane steps at the ends of the x-chunk)
    int t = 0;
    SK_T(6)
    {   // plane xa - 1 reaches only output plane xa (tap d = 0)
        const int lo = issue_upto(t + NSLOT);
        if (xa > 0) single_step(T_{}, F_{}, F_{}, 0, P1, P1, P1);
        SK_T(7)
        end_step(0, lo, issued);
        t = 1;
    }
    const int npairs = n >> 1;
    typedef std::integral_constant<int, 0> S0_;
    typedef std::integral_constant<int, 1> S1_;
    typedef std::integral_constant<int, 2> S2_;
    typedef std::integral_constant<int, 3> S3_;
    // Pair steps alternate between two fixed role assignments: input planes t = 1 + 2 pi live in slots (1, 2) for even pi
    // and (3, 0) for odd pi, and the accumulator pairs P / Q swap roles -- two straight-line bodies, every register fixed.
    auto even_step = [&]() {
        const int lo = issue_upto(t + NSLOT);
        const int xA = xa - 1 + t;   // input plane A = the output plane of the second-oldest accumulator
        SK_T(2)
        pair_step(S1_{}, S2_{}, P0, P1, Q0, Q1);
        SK_T(0)
        int ns = 0;
        if (xA - 1 >= xa) ns += finish_plane(P0, xA - 1);
        ns += finish_plane(P1, xA);
        reset(P0);
        reset(P1);
        end_step(ns, lo, issued);
        t += 2;
    };
    auto odd_step = [&]() {
        const int lo = issue_upto(t + NSLOT);
        const int xA = xa - 1 + t;
        SK_T(2)
        pair_step(S3_{}, S0_{}, Q0, Q1, P0, P1);
        SK_T(0)
        int ns = finish_plane(Q0, xA - 1);
        ns += finish_plane(Q1, xA);
        reset(Q0);
        reset(Q1);
        end_step(ns, lo, issued);
        t += 2;
    };
    for (int pi = 0; pi + 1 < npairs; pi += 2) {
        even_step();
        odd_step();
    }
    if (npairs & 1) even_step();
    if (npairs & 1) {   // the two oldest live planes are Q0 / Q1: rename (once per workgroup)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                P0[i][j] = Q0[i][j];
                P1[i][j] = Q1[i][j];
            }
    }
    if (n & 1) {   // plane xb - 1 alone: taps d = 2 -> P0 (xb - 2), d = 1 -> P1 (xb - 1)
        const int lo = issue_upto(t + NSLOT);
        single_step(F_{}, T_{}, T_{}, t & 3, P0, P1, P0);
        SK_T(7)
        int ns = 0;
        if (xb - 2 >= xa) ns += finish_plane(P0, xb - 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) P0[i][j] = P1[i][j];
        end_step(ns, lo, issued);
        t += 1;
    }
    // plane xb reaches only output plane xb - 1 (tap d = 2)
    if (xb < a.Xt) single_step(F_{}, F_{}, T_{}, t & 3, P0, P0, P0);
    SK_T(7)
    finish_plane(P0, xb - 1);
    SK_T(3)
    SK_T_DUMP(a, w, lane)

    // ---- block-level reduction of the GroupNorm partials (conv3_m16_kernel's) --------------------------------
    if (a.partial) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);
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
            float tsum = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) tsum += red[q * 16 + tid];
            a.partial[((long long)b * nblk + block_in_batch) * 16 + tid] = tsum;
        }
    }
}

// ------------------------------------------------------------------------------------------
// conv3_pxm_kernel (round 4): conv3_px_kernel's plane streaming for precision "mix8" -- the single-chunk 32 -> 32 convs of the
// production tile on [hi fp16 | x8 | lo8] lines.  conv3_m16_kernel's mix8 form stages six planes of the hi halves and six planes of
// the 8-bit halves for four output planes (3.4 plane-halves per output plane-half with the patch halo; it fetches 2.9 x its input
// and is bound by that).  Here every input plane is staged ONCE: a step consumes two input planes in two
// phases -- the fp16 product on their hi halves (conv3_px_kernel's body), the block-scaled fp8 product on their 8-bit halves
// (conv3_m16_kernel's K = 128 blocks) -- and four 11 KiB slots suffice: the 8-bit halves of a step land during its fp16 phase,
// the hi halves of the NEXT step during its fp8 phase, in the slots the fp16 phase has just freed.  Same LDS budget as
// conv3_px_kernel (four slots + twelve half tap rows of fp16 weights: 80 KiB, two workgroups per CU); the 60 KiB fp8 weight image
// streams from L2.  Output: a RAW split pair, statistics from the fp32 accumulators, store box honoured.
template <int RESH, int NPOSP>
__global__ void __launch_bounds__(256, 2) conv3_pxm_kernel(Conv3Args a) {
    constexpr int NSLOT = 4;
    constexpr int plane_bytes = NPOSP * kPosBytes;   // compile-time: a slot's offset is an immediate of its ds_read_b128
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // column tile of this wave (32 voxels of the 128-voxel patch)
    const int c16 = lane & 15, g = lane >> 4;

    int blk = blockIdx.x;   // XCD-aware order, see conv3_kernel
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

    // ---- patch geometry (conv3_m16_kernel's, one column tile per wave) ---------------------------
    const int pitch = a.pitch;
    int off, ybase, zbase, q_row, out_vox0, tile_nvox;
    int svy, svz;             // (y, z) of the voxel this lane STORES: column c16 + 16 (g & 1) of the wave's tile
    unsigned vflags = 0;      // bit j: voxel 16 j + c16 on the z = 0 face | << 8: on the z = Zt-1 face | << 16: inside the tile
    auto zlo = [&](int j) { return (vflags >> j) & 1u; };
    auto zhi = [&](int j) { return (vflags >> (8 + j)) & 1u; };
    auto vvalid = [&](int j) { return (vflags >> (16 + j)) & 1u; };
    // linear mode only (Zt <= 40, conv3_m16_kernel's comment): region position q <-> in-plane voxel v0 - Zt - 1 + q
    const int needed = kPatch + 2 * a.Zt + 2;   // positions a plane really holds; NPOSP rounds it up to a DMA granule
    {
        const int v0 = patch * kPatch;
        off = v0 - a.Zt - 1;
        ybase = 0;
        zbase = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int v = v0 + 32 * w + 16 * j + c16;
            const int vy = v / a.Zt, vz = v - vy * a.Zt;
            vflags |= (unsigned)(v < a.Yt * a.Zt) << (16 + j);
            vflags |= (unsigned)(vz == 0) << j;
            vflags |= (unsigned)(vz == a.Zt - 1) << (8 + j);
            if (j == 0) q_row = v - off;
        }
        out_vox0 = v0 + 32 * w;
        tile_nvox = a.Yt * a.Zt;
        const int sv = out_vox0 + c16 + 16 * (g & 1);
        svy = sv / a.Zt;
        svz = sv - svy * a.Zt;
    }
    const bool sbox = !a.has_box || (svy >= a.box_lo[1] && svy < a.box_hi[1] && svz >= a.box_lo[2] && svz < a.box_hi[2]);

    // ---- LDS-DMA bookkeeping: this lane's slots of a plane (conv3_m16_kernel's swizzle) -----------
    constexpr int ndma = NPOSP / 16;
    int d_vox[kMaxDma];
    const int d_cs = ((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16;
#pragma unroll
    for (int k = 0; k < kMaxDma; ++k) {
        const int t = w + 4 * k;
        const int q = (64 * t + lane) >> 2;
        const int Pq = q + off;
        const int y = ybase + (Pq >= 0 ? Pq / pitch : -1), z = zbase + (Pq >= 0 ? Pq % pitch : 0);
        // positions >= needed are padding that the LDS-DMA never writes (d_vox -2: the lane sits out of the instruction):
        // two of them hold the bias / the GroupNorm coefficients of a raw source, the last four are the zero window
        const bool ok = (t < ndma) && y >= 0 && y < a.Yt && z >= 0 && z < a.Zt;
        d_vox[k] = q >= needed ? -2 : (ok ? (y * a.Zt + z) * 128 + d_cs : -1);   // the BYTE offset of this lane's 16-byte piece of the hi halves in a plane of [hi | x8 | lo8] lines
    }

    const int xa = xc * a.XC;
    const int xb = min(xa + a.XC, a.Xt);
    const int n = xb - xa;                 // output planes xa .. xb-1 of this workgroup
    // The last four positions of every slot are a 256-byte window of zeros (never written by the LDS-DMA).  A tap that
    // leaves the tile through a z face reads zeros at `zero_addr + (its own address & 255)`: the SAME banks its data
    // read would have used.  With one shared zero line instead, the four lanes of a z-face voxel (one per K group, one
    // in each ds_read_b128 lane group) each collide with another lane's banks: 8 LDS cycles instead of 4 for the reads
    // of six of the nine tap rows, SQ_LDS_BANK_CONFLICT 18-35 % of the LDS-active cycles in every conv3 kernel
    // (simulated on the documented lane groups: 6.13 cycles per B read on average against 4.00).
    constexpr int zero_addr = (NPOSP - 4) * kPosBytes;
    static_assert(zero_addr % 256 == 0, "the zero window must cover the 64 banks once");
    constexpr int kOvs = 128;              // bytes per output voxel: a split pair [hi (32) | lo (32)]
    const long long out_plane = (long long)a.Yt * a.Zt * kOvs;
    char* outb = a.out + (long long)b * a.Xt * out_plane;
    const SrcDev s0 = a.src[0];
    const char* srcb = s0.data + (long long)b * s0.batch;

    // H slots 0, 1: the hi halves of the step's two input planes; E slots 2, 3: their 8-bit halves ([x8 | lo8], 64 bytes on in the
    // voxel line).  A plane outside the tile stages zeros (every lane's offset out of range).
    auto issue_half = [&](int slot, int x, int half) {
        const bool xok = x >= 0 && x < a.Xt;
        const __amdgpu_buffer_rsrc_t rsrc = sk::make_rsrc(srcb + (long long)(xok ? x : 0) * s0.plane, (unsigned)s0.plane);
        char* lbase = lds + slot * plane_bytes;
#pragma unroll
        for (int k = 0; k < kMaxDma; ++k) {
            const int tt = w + 4 * k;
            if (tt < ndma && d_vox[k] != -2) {   // padding lanes are masked out of the plane's last piece (EXEC)
                const unsigned voff = (xok && d_vox[k] >= 0) ? (unsigned)(d_vox[k] + 64 * half) : sk::kOob;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lbase + tt * 1024), 16, voff, 0, 0, 0);
            }
        }
    };
    // ---- weights: half rows (tap row dydz, cout half i) 0 .. RESH-1 in registers, RESH .. 17 in LDS behind the ring ----
    // fragment of (row dydz, cout half i, x tap d): ((dydz * 2 + i) * 3 + d) KiB into the packed weight
    const __amdgpu_buffer_rsrc_t wrsrc = sk::make_rsrc(a.wpk, (unsigned)a.wpk_bytes);
    half8 wres[RESH][3];
#pragma unroll
    for (int r = 0; r < RESH; ++r)
#pragma unroll
        for (int d = 0; d < 3; ++d)
            wres[r][d] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16, (r * 3 + d) * 1024, 0));
    char* wlds = lds + NSLOT * plane_bytes;
    for (int i = tid; i < (18 - RESH) * 3 * 64; i += 256)
        *reinterpret_cast<uint4*>(wlds + i * 16) = *reinterpret_cast<const uint4*>(a.wpk + RESH * 3 * 1024 + i * 16);
    // padding positions of a slot: needed, needed + 1 (128 bytes: slot 0 the GroupNorm scales of a raw source, slot 1 the
    // bias, slot 2 the GroupNorm shifts) | the zero window NPOSP - 4 .. NPOSP - 1
    if (tid < NSLOT * 16)
        *reinterpret_cast<uint4*>(lds + (tid >> 4) * plane_bytes + zero_addr + (tid & 15) * 16) = make_uint4(0, 0, 0, 0);
    if (tid >= 128 && tid < 160) reinterpret_cast<float*>(lds + plane_bytes + needed * kPosBytes)[tid - 128] = a.bias[tid - 128];
    SK_T_DECL
    issue_half(0, xa - 1, 0);
    issue_half(1, xa, 0);
    issue_half(2, xa - 1, 1);
    issue_half(3, xa, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- accumulators ----------------------------------------------------------------------------------------
    // element r of [i][j]: cout 16 i + 4 g + r, voxel 16 j + c16 of the wave's column tile
    f32x4 P0[2][2], P1[2][2], Q0[2][2], Q1[2][2];
    // the bias (the accumulators' initial value) is re-read from its copy in LDS -- padding positions of slot 1 -- at every
    // reset: eight registers less in the loop
    const float* lbias = reinterpret_cast<const float*>(lds + plane_bytes + needed * kPosBytes) + 4 * g;
    auto reset = [&](f32x4 (&o)[2][2]) {
        o[0][0] = o[0][1] = *reinterpret_cast<const f32x4*>(lbias);
        o[1][0] = o[1][1] = *reinterpret_cast<const f32x4*>(lbias + 16);
    };
    reset(P0);   // (behind the barrier above: the bias in LDS was written by another wave)
    reset(P1);
    reset(Q0);
    reset(Q1);
    float gsum[2] = {0.0f, 0.0f}, gsq[2] = {0.0f, 0.0f};

    auto baddr = [&](int dydz, int j) -> int {
        const int dz = dydz % 3 - 1;
        const int q = q_row + (dydz / 3 - 1) * pitch + dz;
        int addr = (q * 4 + (g ^ (((q >> 2) & 1) << 1))) * 16 + 1024 * j;
        if (dz < 0) addr = zlo(j) ? zero_addr + (addr & 255) : addr;
        if (dz > 0) addr = zhi(j) ? zero_addr + (addr & 255) : addr;
        return addr;
    };
    auto wfrag = [&](int dydz, int i, half8 (&dst)[3]) {
        if (2 * dydz + i < RESH) {
#pragma unroll
            for (int d = 0; d < 3; ++d) dst[d] = wres[2 * dydz + i][d];
        } else {
            const char* p = wlds + (2 * dydz + i - RESH) * 3 * 1024 + lane * 16;
#pragma unroll
            for (int d = 0; d < 3; ++d) dst[d] = *reinterpret_cast<const half8*>(p + d * 1024);
        }
    };

    // A step over the input planes A (slot sA) and B = A + 1 (slot sB).  oA1 / oA / oB / oB1: the accumulators of the
    // output planes A-1, A, B, B+1.  Tap d of a weight row multiplies x_in = x_out + d - 1.
    auto pair_step = [&](auto SA, auto SB, f32x4 (&oA1)[2][2], f32x4 (&oA)[2][2], f32x4 (&oB)[2][2], f32x4 (&oB1)[2][2]) {
        // compile-time slots: the 18 tap addresses of the patch (plane-relative, loop-invariant) serve both planes of
        // every step through the immediate offset of ds_read_b128
        const char* pa = lds + decltype(SA)::value * plane_bytes;
        const char* pb = lds + decltype(SB)::value * plane_bytes;
        half8 bq[2][2][2];   // [buffer][plane][j]: the B fragments of a tap row, one row ahead
        half8 wq[2][3];      // [buffer][d]: the weight fragments of a half row (cout half i of a tap row), one half row ahead
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ad = baddr(0, j);
            bq[0][0][j] = *reinterpret_cast<const half8*>(pa + ad);
            bq[0][1][j] = *reinterpret_cast<const half8*>(pb + ad);
        }
        wfrag(0, 0, wq[0]);
        // The 18 half rows of a step, 12 MFMAs each (192 cycles of the matrix pipe).  The LDS reads of half row h + 1 -- its
        // three weight fragments when its tap row lives in LDS, and the four B fragments of the next tap row -- are issued
        // in the FIRST MFMA gaps of half row h, in the order half row h + 1 consumes them, so the youngest read is 80+
        // cycles old (and not needed before the seventh MFMA) when half row h + 1 starts.  The order is pinned: left to
        // itself the scheduler spread the reads to the END of the half row and every half row began with an
        // `s_waitcnt lgkmcnt(0)` on a read issued one MFMA earlier (SQ_WAIT_ANY 38 % of the wave-cycles).
#pragma unroll
        for (int dydz = 0; dydz < 9; ++dydz) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int h = dydz * 2 + i;   // its weights sit in wq[h & 1], its B fragments in bq[dydz & 1]
                const half8(&W)[3] = wq[h & 1];
                const int ndy = i == 0 ? dydz : dydz + 1, ni = i ^ 1;         // the next half row
                bool wread = ndy < 9 && 2 * ndy + ni >= RESH;                  // ... reads its weights from LDS
                bool bread = i == 1 && dydz < 8;                               // ... starts a new tap row: B fragments
                if (SK_PX_ABL(16) && wread) {   // timing experiment: no LDS weight reads (a resident half row instead)
                    wread = false;
                    wfrag((2 * ndy + ni) % RESH / 2, (2 * ndy + ni) % RESH % 2, wq[(h + 1) & 1]);
                } else if (ndy < 9 && !wread) {
                    wfrag(ndy, ni, wq[(h + 1) & 1]);                           // resident row: register names only
                }
                if (SK_PX_ABL(32) && bread) {   // timing experiment: no B fragment reads after the first tap row
                    bread = false;
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) bq[(dydz + 1) & 1][pl][jj] = bq[dydz & 1][pl][jj];
                }
                // read k of the next half row, in consumption order: W0, A0, B0, W1, W2, A1, B1
                auto next_read = [&](int k) {
                    const char* wp = wlds + (2 * ndy + ni - RESH) * 3 * 1024 + lane * 16;
                    int kk = k;
                    if (!wread) kk = (k == 0 ? 1 : k == 1 ? 2 : k == 2 ? 5 : 6);   // B fragments only: A0, B0, A1, B1
                    if (!bread && kk > 0) kk = (kk == 1 ? 3 : 4);                  // weights only: W0, W1, W2
                    switch (kk) {
                        case 0: wq[(h + 1) & 1][0] = *reinterpret_cast<const half8*>(wp); break;
                        case 1: bq[(dydz + 1) & 1][0][0] = *reinterpret_cast<const half8*>(pa + baddr(dydz + 1, 0)); break;
                        case 2: bq[(dydz + 1) & 1][1][0] = *reinterpret_cast<const half8*>(pb + baddr(dydz + 1, 0)); break;
                        case 3: wq[(h + 1) & 1][1] = *reinterpret_cast<const half8*>(wp + 1024); break;
                        case 4: wq[(h + 1) & 1][2] = *reinterpret_cast<const half8*>(wp + 2048); break;
                        case 5: bq[(dydz + 1) & 1][0][1] = *reinterpret_cast<const half8*>(pa + baddr(dydz + 1, 1)); break;
                        default: bq[(dydz + 1) & 1][1][1] = *reinterpret_cast<const half8*>(pb + baddr(dydz + 1, 1)); break;
                    }
                };
                const int nreads = (wread ? 3 : 0) + (bread ? 4 : 0);
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    const int j = m / 6;
                    const half8 fa = bq[dydz & 1][0][j], fb = bq[dydz & 1][1][j];
                    switch (m % 6) {   // tap d of a weight row multiplies x_in = x_out + d - 1
                        case 0: oB[i][j] = SK_MFMA_16x16x32_T16(W[0], fa, oB[i][j], 0, 0, 0); break;
                        case 1: oB1[i][j] = SK_MFMA_16x16x32_T16(W[0], fb, oB1[i][j], 0, 0, 0); break;
                        case 2: oA[i][j] = SK_MFMA_16x16x32_T16(W[1], fa, oA[i][j], 0, 0, 0); break;
                        case 3: oA1[i][j] = SK_MFMA_16x16x32_T16(W[2], fa, oA1[i][j], 0, 0, 0); break;
                        case 4: oB[i][j] = SK_MFMA_16x16x32_T16(W[1], fb, oB[i][j], 0, 0, 0); break;
                        default: oA[i][j] = SK_MFMA_16x16x32_T16(W[2], fb, oA[i][j], 0, 0, 0); break;
                    }
                    if (m < nreads) next_read(m);
                    if (m <= nreads) __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // The fp8 phase of a step: the same two input planes' 8-bit halves (E slots 2, 3) against the fp8 weight image, tap rows in
    // pairs (conv3_m16_kernel's K = 128 blocks: lane group g >> 1 = the row of the pair, g & 1 = x8 | lo8).
    auto pair_f8 = [&](f32x4 (&oA1)[2][2], f32x4 (&oA)[2][2], f32x4 (&oB)[2][2], f32x4 (&oB1)[2][2]) {
        typedef int v8i __attribute__((ext_vector_type(8)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const char* pa = lds + 2 * plane_bytes;
        const char* pb = lds + 3 * plane_bytes;
        const int halfsel = (g & 1) * 2;
        const int sa = a.w8_scale, sb = 0x70707070;   // E8M0: weights 2^-b (host), activations 2^-15
        const unsigned w8 = (unsigned)a.w8_off + (unsigned)lane * 32u;
        auto rd = [&](const char* p) {
            const u32x4 lo4 = *reinterpret_cast<const u32x4*>(p);
            const u32x4 hi4 = *reinterpret_cast<const u32x4*>(p + 16);
            return v8i{(int)lo4[0], (int)lo4[1], (int)lo4[2], (int)lo4[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
        };
#pragma unroll 1
        for (int rp = 0; rp < 5; ++rp) {
            const int rr = min(2 * rp + (g >> 1), 8);   // (the tenth row does not exist: its weights are zero)
            const int dz = rr % 3 - 1;
            const int q = q_row + (rr / 3 - 1) * pitch + dz;
            v8i af[2][3];
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const unsigned wo = w8 + (unsigned)(((rp * 2 + i2) * 3 + d) * 2048);
                    const u32x4 w0 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wo, 0, 0);
                    const u32x4 w1 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wo + 16, 0, 0);
                    af[i2][d] = v8i{(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
                }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int addr = (q * 4 + (halfsel ^ (((q >> 2) & 1) << 1))) * 16 + 1024 * j;
                if ((dz < 0 && zlo(j)) || (dz > 0 && zhi(j))) addr = zero_addr + (addr & 255);
                const v8i fa = rd(pa + addr), fb = rd(pb + addr);
#pragma unroll
                for (int i = 0; i < 2; ++i) {   // tap d of a weight row multiplies x_in = x_out + d - 1
                    oB[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i][0], fa, oB[i][j], 0, 0, 0, sa, 0, sb);
                    oB1[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i][0], fb, oB1[i][j], 0, 0, 0, sa, 0, sb);
                    oA[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i][1], fa, oA[i][j], 0, 0, 0, sa, 0, sb);
                    oA1[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i][2], fa, oA1[i][j], 0, 0, 0, sa, 0, sb);
                    oB[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i][1], fb, oB[i][j], 0, 0, 0, sa, 0, sb);
                    oA[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i][2], fb, oA[i][j], 0, 0, 0, sa, 0, sb);
                }
            }
        }
    };

    // store of one finished output plane as a RAW split pair [hi | lo] + its GroupNorm partial sums from the fp32 accumulators
    // (conv3_m16_kernel's SPLIT epilogue); returns the number of store instructions issued (0 or 4)
    auto finish_plane = [&](f32x4 (&o)[2][2], int x) -> int {
        const bool xbox = !a.has_box || (x >= a.box_lo[0] && x < a.box_hi[0]);   // wave-uniform
        const __amdgpu_buffer_rsrc_t rout = sk::make_rsrc(outb + (long long)x * out_plane, (unsigned)out_plane);
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                unsigned d[2][2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 r = o[i][j];
                    half4 hv = {(t16)r[0], (t16)r[1], (t16)r[2], (t16)r[3]};
                    if (part == 1)   // lo = fp16(v - hi): exact difference, rounded once
                        hv = half4{(t16)(r[0] - (float)hv[0]), (t16)(r[1] - (float)hv[1]), (t16)(r[2] - (float)hv[2]), (t16)(r[3] - (float)hv[3])};
                    const uint2 u = __builtin_bit_cast(uint2, hv);
                    d[j][0] = u.x;
                    d[j][1] = u.y;
                    if (part == 0 && vvalid(j)) {
                        gsum[i] += (r[0] + r[1]) + (r[2] + r[3]);
                        gsq[i] += (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
                    }
                }
                if (xbox) {
                    const auto s0_ = __builtin_amdgcn_permlane16_swap(d[0][0], d[1][0], false, false);
                    const auto s1_ = __builtin_amdgcn_permlane16_swap(d[0][1], d[1][1], false, false);
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 lv = {s0_[0], s1_[0], s0_[1], s1_[1]};
                    const int vv = c16 + 16 * (g & 1);
                    const bool sok = sbox && out_vox0 + vv < tile_nvox;
                    const unsigned so = (unsigned)((out_vox0 + vv) * kOvs + 64 * part + 32 * i + 16 * (g >> 1));
                    __builtin_amdgcn_raw_buffer_store_b128(lv, rout, sok ? so : sk::kOob, 0, 0);
                }
            }
        return xbox ? 4 : 0;
    };
    typedef std::integral_constant<int, 0> S0_;
    typedef std::integral_constant<int, 1> S1_;
    // ---- the march -----------------------------------------------------------------------------------------------
    // Step s consumes the input planes A = xa - 1 + 2 s and B = A + 1 in two phases and completes the output planes A - 1
    // and A.  Every input plane xa - 1 .. xb goes through a pair step (planes outside the tile are zeros; the taps of the
    // end planes that reach outside [xa, xb) land in accumulators that are never stored: ~1 % of an x-chunk's MFMAs).
    //   fp16 phase on the H slots | wait: the 8-bit halves have landed | barrier: every wave is done with the H slots |
    //   LDS-DMA of the NEXT step's hi halves into the H slots | fp8 phase on the E slots | epilogue | counted wait: the
    //   hi halves have landed | barrier: every wave is done with the E slots | LDS-DMA of the next step's 8-bit halves
    // -- each LDS-DMA has a whole phase of matrix instructions to land in.
    const int nsteps = (n + 3) >> 1;
    auto step = [&](int s, f32x4 (&oA1)[2][2], f32x4 (&oA)[2][2], f32x4 (&oB)[2][2], f32x4 (&oB1)[2][2]) {
        const int xA = xa - 1 + 2 * s;
        pair_step(S0_{}, S1_{}, oA1, oA, oB, oB1);
        SK_T(0)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        SK_T(1)
        if (s + 1 < nsteps) {
            issue_half(0, xA + 2, 0);
            issue_half(1, xA + 3, 0);
        }
        SK_T(2)
        pair_f8(oA1, oA, oB, oB1);
        SK_T(0)
        int ns = 0;
        if (xA - 1 >= xa && xA - 1 < xb) ns += finish_plane(oA1, xA - 1);
        if (xA >= xa && xA < xb) ns += finish_plane(oA, xA);
        reset(oA1);
        reset(oA);
        SK_T(3)
        if (ns == 8)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ns == 4)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SK_T(4)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        SK_T(1)
        if (s + 1 < nsteps) {
            issue_half(2, xA + 2, 1);
            issue_half(3, xA + 3, 1);
        }
        SK_T(2)
    };
    SK_T(6)
    for (int s = 0; s < nsteps; s += 2) {   // the accumulator pairs P / Q swap roles: two straight-line bodies
        step(s, P0, P1, Q0, Q1);
        if (s + 1 < nsteps) step(s + 1, Q0, Q1, P0, P1);
    }
    SK_T_DUMP(a, w, lane)

    // ---- block-level reduction of the GroupNorm partials (conv3_m16_kernel's) --------------------------------
    if (a.partial) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);
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
            float tsum = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) tsum += red[q * 16 + tid];
            a.partial[((long long)b * nblk + block_in_batch) * 16 + tid] = tsum;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Gather GEMM for the layers with no spatial reuse of activations: 2x2x2 stride-2
// down-sampling convs (8 taps, each input voxel feeds one output voxel) and 1x1x1
// channel reducers.  B fragments come straight from global memory (16 B per lane),
// weights in the same packed A-fragment order.  A wave owns PV tiles of 32 voxels.
// ------------------------------------------------------------------------------------------
struct GatherArgs {
    const char* in;
    const float* affine;    // (B, 2, Cin) fp32 or NULL: input is RAW, apply silu(a*x + b) on load
    const char* wpk;
    const float* bias;
    char* out;
    float* partial;
    int B, Xo, Yo, Zo;      // output extents
    int Xi, Yi, Zi, Cin;    // input extents / channels
    int ksize;              // 1 or 2 (stride == ksize)
    int nblk;
};

// SPLIT: input and output voxel lines are [hi | lo] fp16 pairs (value = hi + lo), the weights come as hi fragments
// followed by lo fragments; per K step w_lo*x_hi + w_hi*x_lo + w_hi*x_hi (the dropped lo*lo term is ~2^-22 relative).
template <int COUT, int PV, int D, bool SPLIT = false>
__global__ void __launch_bounds__(256) gather_gemm_kernel(GatherArgs a) {
    constexpr int NT = COUT / 32;
    constexpr int kOvs = COUT * 2 * (SPLIT ? 2 : 1);   // bytes per output voxel
    const long long lstride = (long long)a.Cin * 2 * (SPLIT ? 2 : 1);   // bytes per input voxel
    __shared__ float red[4 * NT * 16];
    __shared__ float aff[2 * 128];   // silu(a*x + b) coefficients of this batch item (RAW input only)
    __shared__ __attribute__((aligned(16))) char gpad[4 * kPadBytes];   // per-wave transpose pads of the epilogue
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / a.nblk;
    const int blk = blockIdx.x % a.nblk;
    const long long nvox = (long long)a.Xo * a.Yo * a.Zo;
    const int ntaps = a.ksize * a.ksize * a.ksize;
    const int nks = a.Cin / 16;
    const int nsteps = ntaps * nks;      // K steps of 16 channels; host guarantees nsteps % D == 0

    f32x16 acc[PV][NT];
    long long vin[PV];   // input voxel index of tap (0,0,0)
    bool ok[PV];
#pragma unroll
    for (int p = 0; p < PV; ++p) {
        long long v = ((long long)blk * 4 + w) * (32 * PV) + p * 32 + col;
        ok[p] = v < nvox;
        long long vv = ok[p] ? v : 0;
        int z = (int)(vv % a.Zo);
        long long t = vv / a.Zo;
        int y = (int)(t % a.Yo), x = (int)(t / a.Yo);
        vin[p] = ((long long)(x * a.ksize) * a.Yi + y * a.ksize) * a.Zi + z * a.ksize;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + 32 * nt + 8 * q + 4 * h);
                acc[p][nt][4 * q] = bv[0];
                acc[p][nt][4 * q + 1] = bv[1];
                acc[p][nt][4 * q + 2] = bv[2];
                acc[p][nt][4 * q + 3] = bv[3];
            }
        }
    }
    const bool raw = a.affine != nullptr;
    if (raw) {
        for (int i = tid; i < 2 * a.Cin; i += 256) aff[i] = a.affine[(long long)b * 2 * a.Cin + i];
        __syncthreads();
    }
    const char* inb = a.in + (long long)b * a.Xi * a.Yi * a.Zi * lstride;
    const char* wlo = a.wpk + (long long)nsteps * NT * 1024;   // SPLIT: the lo fragments follow the hi fragments
    // Software pipeline, D steps deep: the B fragments (16 B per lane straight from HBM, no reuse) and the weight
    // fragments of step s + D are requested before step s is multiplied -- a wave keeps D*(PV + NT) KiB in flight
    // instead of one load -> wait -> MFMA round trip per step.
    half8 afr[D][NT], bfr[D][PV];
    half8 alo[SPLIT ? D : 1][NT], blo[SPLIT ? D : 1][PV];
#define SK_GATHER_ISSUE(S, SLOT)                                                                              \
    {                                                                                                         \
        const int s_ = (S);                                                                                   \
        const int tap_ = s_ / nks, ks_ = s_ - tap_ * nks;                                                     \
        const int dx_ = tap_ / (a.ksize * a.ksize), dy_ = (tap_ / a.ksize) % a.ksize, dz_ = tap_ % a.ksize;   \
        const long long toff_ = ((long long)dx_ * a.Yi + dy_) * a.Zi + dz_;                                   \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                   \
            afr[SLOT][nt] = *reinterpret_cast<const half8*>(a.wpk + ((long long)s_ * NT + nt) * 1024 + lane * 16); \
            if constexpr (SPLIT)                                                                              \
                alo[SLOT][nt] = *reinterpret_cast<const half8*>(wlo + ((long long)s_ * NT + nt) * 1024 + lane * 16); \
        }                                                                                                     \
        _Pragma("unroll") for (int p = 0; p < PV; ++p) {                                                      \
            const char* src_ = inb + (vin[p] + toff_) * lstride + (ks_ * 16 + 8 * h) * 2;                     \
            bfr[SLOT][p] = *reinterpret_cast<const half8*>(src_);                                             \
            if constexpr (SPLIT) blo[SLOT][p] = *reinterpret_cast<const half8*>(src_ + a.Cin * 2);            \
        }                                                                                                     \
    }
#pragma unroll
    for (int d = 0; d < D; ++d) SK_GATHER_ISSUE(d, d)
    // no branch around the requests of the steady state: a conditional issue makes the compiler's vmcnt
    // bookkeeping fall back to vmcnt(0) at the join, i.e. one round trip per step again
#define SK_GATHER_STEP(S, SLOT, PREFETCH)                                                                     \
    {                                                                                                         \
        const int s = (S);                                                                                    \
        half8 av[NT], bv[PV], avl[NT], bvl[PV];                                                               \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) av[nt] = afr[SLOT][nt];                             \
        _Pragma("unroll") for (int p = 0; p < PV; ++p) bv[p] = bfr[SLOT][p];                                  \
        if constexpr (SPLIT) {                                                                                \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) avl[nt] = alo[SLOT][nt];                        \
            _Pragma("unroll") for (int p = 0; p < PV; ++p) bvl[p] = blo[SLOT][p];                             \
        }                                                                                                     \
        if (PREFETCH) SK_GATHER_ISSUE(s + D, SLOT)                                                            \
        if (raw) { /* this lane's 8 input channels of the K step */                                           \
            const int c0 = (s % nks) * 16 + 8 * h;                                                            \
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(aff + c0);                                       \
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(aff + c0 + 4);                                   \
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(aff + a.Cin + c0);                               \
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(aff + a.Cin + c0 + 4);                           \
            const float ga[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};                     \
            const float gb[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};                     \
            _Pragma("unroll") for (int p = 0; p < PV; ++p) _Pragma("unroll") for (int j = 0; j < 8; ++j) {    \
                if constexpr (SPLIT) { /* gn_silu_split_kernel's arithmetic, op for op: activate hi + lo, split again */ \
                    const float y = fmaf(ga[j], (float)bv[p][j] + (float)bvl[p][j], gb[j]);                   \
                    const float sv = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));                            \
                    bv[p][j] = sk::round_t16(sv);                                                             \
                    bvl[p][j] = (t16)(sv - (float)bv[p][j]);                                                  \
                } else {                                                                                      \
                    float y = fmaf(ga[j], (float)bv[p][j], gb[j]);                                            \
                    bv[p][j] = sk::round_t16(y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)));                   \
                }                                                                                             \
            }                                                                                                 \
        }                                                                                                     \
        _Pragma("unroll") for (int p = 0; p < PV; ++p) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {    \
            if constexpr (SPLIT) {                                                                            \
                acc[p][nt] = SK_MFMA_32x32x16_T16(avl[nt], bv[p], acc[p][nt], 0, 0, 0);     \
                acc[p][nt] = SK_MFMA_32x32x16_T16(av[nt], bvl[p], acc[p][nt], 0, 0, 0);     \
            }                                                                                                 \
            acc[p][nt] = SK_MFMA_32x32x16_T16(av[nt], bv[p], acc[p][nt], 0, 0, 0);          \
        }                                                                                                     \
    }
    int s0 = 0;
    for (; s0 + D < nsteps; s0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) SK_GATHER_STEP(s0 + d, d, true)
    }
#pragma unroll
    for (int d = 0; d < D; ++d) SK_GATHER_STEP(s0 + d, d, false)
#undef SK_GATHER_STEP
#undef SK_GATHER_ISSUE
    float gsum[NT][4], gsq[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) gsum[nt][q] = gsq[nt][q] = 0.0f;
    char* outb = a.out + (long long)b * nvox * kOvs;
    // Each 32 x 32 (channel, voxel) tile leaves the MFMA layout through a per-wave LDS pad as whole 64-byte pieces of
    // the voxel lines, 16 B per lane (the tile's 32 voxels are consecutive in the output): 8-byte stores straight from
    // the accumulator layout write every 64-B piece in four partial transactions.
    char* pad = gpad + w * kPadBytes;
    const int rv = lane >> 2, rc = lane & 3;
    const long long vbase = ((long long)blk * 4 + w) * (32 * PV);   // first voxel of this wave
#pragma unroll
    for (int p = 0; p < PV; ++p) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int part = 0; part < (SPLIT ? 2 : 1); ++part) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v0 = acc[p][nt][4 * q], v1 = acc[p][nt][4 * q + 1];
                    float v2 = acc[p][nt][4 * q + 2], v3 = acc[p][nt][4 * q + 3];
                    half4 hv = {(t16)v0, (t16)v1, (t16)v2, (t16)v3};
                    if (part == 1)
                        hv = half4{(t16)(v0 - (float)hv[0]), (t16)(v1 - (float)hv[1]), (t16)(v2 - (float)hv[2]),
                                   (t16)(v3 - (float)hv[3])};
                    *reinterpret_cast<half4*>(pad + col * kPadStride + ((q ^ ((col >> 1) & 3)) * 16) + 8 * h) = hv;
                    if (part == 0 && ok[p]) {
                        gsum[nt][q] += (v0 + v1) + (v2 + v3);
                        gsq[nt][q] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
                    }
                }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int vv = rv + 16 * hh;
                    const half8 line = *reinterpret_cast<const half8*>(pad + vv * kPadStride + ((rc ^ ((vv >> 1) & 3)) * 16));
                    const long long v = vbase + p * 32 + vv;
                    if (v < nvox)
                        *reinterpret_cast<half8*>(outb + v * kOvs + part * (COUT * 2) + nt * 64 + rc * 16) = line;
                }
            }
        }
    }
    if (a.partial) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float s = gsum[nt][q], ss = gsq[nt][q];
#pragma unroll
                for (int m = 16; m > 0; m >>= 1) {
                    s += __shfl_xor(s, m);
                    ss += __shfl_xor(ss, m);
                }
                if (col == 0) {
                    int quad = 8 * nt + 2 * q + h;
                    red[(w * (NT * 8) + quad) * 2 + 0] = s;
                    red[(w * (NT * 8) + quad) * 2 + 1] = ss;
                }
            }
        __syncthreads();
        if (tid < NT * 16) {
            float t = red[tid] + red[NT * 16 + tid] + red[2 * NT * 16 + tid] + red[3 * NT * 16 + tid];
            a.partial[((long long)b * a.nblk + blk) * (NT * 16) + tid] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------
// 2x2x2 stride-2 down conv, LDS-staged, with the producer's GroupNorm + SiLU folded in.
//
// Every input voxel of a stride-2 conv feeds exactly ONE output voxel, so the kernel that reads the input for the
// conv can also be the pass that activates it: the RAW tensor (+ the affine of its GroupNorm) comes in, each
// workgroup stages the input of its output voxels through LDS by LDS-DMA -- whole 128-byte lines per 8 lanes, where
// gather_gemm_kernel's per-lane 16-byte loads touch 32 lines per instruction -- activates the staged bytes in place
// in LDS (the same arithmetic as gn_silu_kernel: bit-identical values), writes them back to the tensor (its other
// consumer, the decoder's skip conv, needs it activated) and multiplies from LDS.  Replaces the separate in-place
// GroupNorm pass over the skip tensor (read + write of the whole tensor) plus the gather kernel's read.
//
// A workgroup owns NV = 8192 / CIN consecutive output voxels.  Stage (dx, dy): the two z-adjacent input voxels
// (dz = 0, 1) of every output voxel are one contiguous row of RB = 4 CIN bytes; stage buffer = NV rows = 32 KiB,
// double buffered.  16-byte chunk c of row n sits at chunk slot c ^ g(n) (g: see below) so that every ds_read_b128
// lane group of the B-fragment reads covers all 64 banks; the swizzle is applied on the DMA's SOURCE address (the
// destination of an LDS-DMA is lane-linear).  Thread tid stages, activates and writes back slots tid + 256 j: its
// chunk index is the same for every j, so its 8 + 8 coefficients live in registers.
// ------------------------------------------------------------------------------------------
struct DownArgs {
    char* in;               // (B, Xi, Yi, Zi, CIN) fp16: raw (affine != NULL) or activated
    const float* affine;    // (B, 2, CIN) or NULL
    const char* wpk;        // [tap (dx,dy,dz)][ks][nt] fragments (sk_conv3d_pack_weight_host, ksize 2)
    const float* bias;
    char* out;              // (B, Xo, Yo, Zo, COUT) fp16 raw
    float* partial;
    const char* zeros;
    int B, Xo, Yo, Zo;
    int writeback;          // store the activated input back
    int nblk;
};

template <int COUT, int CIN>
__global__ void __launch_bounds__(256, 2) down2_act_kernel(DownArgs a) {
    constexpr int NT = COUT / 32;
    constexpr int RB = 4 * CIN;                // bytes per staged row: two voxels x CIN fp16
    constexpr int CPR = RB / 16;               // 16-byte chunks per row (8 | 16)
    constexpr int NV = 32768 / RB;             // output voxels per workgroup (256 | 128)
    // wave w: cout tiles [NTW * wn, NTW * (wn + 1)), column tiles [PV * wm, PV * (wm + 1)): two of each, so that a weight
    // fragment and a B fragment both feed two MFMAs (COUT 128 x CIN 64 with one column tile per wave streamed four
    // weight fragments per B fragment and spilled)
    constexpr int WN = NT / 2;                 // wave groups along cout (1 | 2)
    constexpr int NTW = NT / WN;               // cout tiles per wave (2)
    constexpr int PV = (NV / 32) / (4 / WN);   // column tiles per wave (2)
    constexpr int NKS = CIN / 16;              // K steps per input voxel
    constexpr int kStage = 32768;
    static_assert(CIN == 32 || CIN == 64, "rows of 128 or 256 bytes");
    extern __shared__ __attribute__((aligned(16))) char dlds[];   // [2][32 KiB] stages + 4 epilogue pads
    __shared__ float red[4 * 2 * 16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int wn = w % WN, wm = w / WN;
    const int b = blockIdx.x / a.nblk, blk = blockIdx.x % a.nblk;
    const int Yi = 2 * a.Yo, Zi = 2 * a.Zo;
    const long long nvox = (long long)a.Xo * a.Yo * a.Zo;
    const long long v0 = (long long)blk * NV;
    // swizzle term of row n: RB 128 -> rows alternate bank halves: (n >> 1) & 7; RB 256 -> every row starts at bank 0: n & 15
    auto gsw = [](int n) { return CPR == 8 ? ((n >> 1) & 7) : (n & 15); };

    // ---- this thread's slots: slot = tid + 256 j  ->  row n = slot / CPR, chunk slot cs = slot % CPR (same for all j)
    const int cs = tid % CPR;
    const int csrc = cs ^ gsw(tid / CPR);      // source chunk: (tid / CPR + (256 / CPR) j) keeps its swizzle term for every j
    long long vin[8];                          // input voxel index of tap (0, 0, 0), or -1 beyond the tensor
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = tid / CPR + (256 / CPR) * j;
        const long long v = v0 + n;
        if (v < nvox) {
            const int zo = (int)(v % a.Zo);
            const long long t = v / a.Zo;
            const int yo = (int)(t % a.Yo), xo = (int)(t / a.Yo);
            vin[j] = ((long long)(2 * xo) * Yi + 2 * yo) * Zi + 2 * zo;
        } else {
            vin[j] = -1;
        }
    }
    char* inb = a.in + (long long)b * 8 * nvox * (CIN * 2);
    const bool raw = a.affine != nullptr;
    float ga[8], gb[8];
    if (raw) {
        const int c0 = (csrc % (CIN / 8)) * 8;   // channels of this thread's chunk
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ga[j] = a.affine[(long long)b * 2 * CIN + c0 + j];
            gb[j] = a.affine[(long long)b * 2 * CIN + CIN + c0 + j];
        }
    }
    auto issue_stage = [&](int st) {
        const long long toff = ((long long)(st >> 1) * Yi + (st & 1)) * Zi;   // (dx, dy)
        char* lbase = dlds + (st & 1) * kStage;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const char* g = vin[j] >= 0 ? inb + (vin[j] + toff) * (CIN * 2) + csrc * 16 : a.zeros + lane * 16;
            dma16(g, lbase + (w + 4 * j) * 1024);
        }
    };

    f32x16 acc[PV][NTW];
#pragma unroll
    for (int p = 0; p < PV; ++p)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + 32 * (NTW * wn + nt) + 8 * q + 4 * h);
                acc[p][nt][4 * q] = bv[0];
                acc[p][nt][4 * q + 1] = bv[1];
                acc[p][nt][4 * q + 2] = bv[2];
                acc[p][nt][4 * q + 3] = bv[3];
            }

    issue_stage(0);
    for (int st = 0; st < 4; ++st) {
        if (st + 1 < 4) {
            issue_stage(st + 1);   // its buffer was last read by stage st - 1's MFMAs: the closing barrier below
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // everything older than those 8 pieces has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // weight fragments of this stage's 2 NKS K steps: requested before the activation pass, used after it
        half8 afr[2 * NKS][NTW];
#pragma unroll
        for (int k = 0; k < 2 * NKS; ++k)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                afr[k][nt] = *reinterpret_cast<const half8*>(a.wpk + ((long long)(st * 2 * NKS + k) * NT + NTW * wn + nt) * 1024 +
                                                             lane * 16);
        char* buf = dlds + (st & 1) * kStage;
        if (raw) {
            const long long toff = ((long long)(st >> 1) * Yi + (st & 1)) * Zi;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                half8* lp = reinterpret_cast<half8*>(buf + (tid + 256 * j) * 16);
                const half8 v = *lp;
                half8 r;
#pragma unroll
                for (int e = 0; e < 8; ++e) {   // gn_silu_kernel's arithmetic, op for op
                    const float y = fmaf(ga[e], (float)v[e], gb[e]);
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-y));
                    r[e] = sk::round_t16(y * sg);
                }
                *lp = r;
                if (a.writeback && vin[j] >= 0)
                    *reinterpret_cast<half8*>(inb + (vin[j] + toff) * (CIN * 2) + csrc * 16) = r;
            }
        }
        __syncthreads();
        // K steps of the stage: (dz, ks); weight step index = ((dx*2 + dy)*2 + dz) * NKS + ks = st * 2 NKS + dz * NKS + ks
#pragma unroll
        for (int k = 0; k < 2 * NKS; ++k) {
#pragma unroll
            for (int p = 0; p < PV; ++p) {
                const int n = 32 * (wm * PV + p) + col;
                const half8 bfr = *reinterpret_cast<const half8*>(buf + n * RB + (((2 * k + h) ^ gsw(n)) * 16));
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    acc[p][nt] = SK_MFMA_32x32x16_T16(afr[k][nt], bfr, acc[p][nt], 0, 0, 0);
            }
        }
        __syncthreads();   // stage buffer free for the DMA of stage st + 2
    }

    // ---- epilogue: transposed 16-byte stores + GroupNorm partials (as gather_gemm_kernel) --------------------
    float gsum[NTW][4], gsq[NTW][4];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) gsum[nt][q] = gsq[nt][q] = 0.0f;
    char* outb = a.out + (long long)b * nvox * (COUT * 2);
    char* pad = dlds + 2 * kStage + w * kPadBytes;
    const int rv = lane >> 2, rc = lane & 3;
    const long long vbase = v0 + (long long)wm * (32 * PV);
#pragma unroll
    for (int p = 0; p < PV; ++p) {
        const bool okp = vbase + p * 32 + col < nvox;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x0 = acc[p][nt][4 * q], x1 = acc[p][nt][4 * q + 1];
                float x2 = acc[p][nt][4 * q + 2], x3 = acc[p][nt][4 * q + 3];
                half4 hv = {(t16)x0, (t16)x1, (t16)x2, (t16)x3};
                *reinterpret_cast<half4*>(pad + col * kPadStride + ((q ^ ((col >> 1) & 3)) * 16) + 8 * h) = hv;
                if (okp) {
                    gsum[nt][q] += (x0 + x1) + (x2 + x3);
                    gsq[nt][q] += (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
                }
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int vv = rv + 16 * hh;
                const half8 line = *reinterpret_cast<const half8*>(pad + vv * kPadStride + ((rc ^ ((vv >> 1) & 3)) * 16));
                const long long v = vbase + p * 32 + vv;
                if (v < nvox) *reinterpret_cast<half8*>(outb + v * (COUT * 2) + (NTW * wn + nt) * 64 + rc * 16) = line;
            }
        }
    }
    if (a.partial) {
        // red[wave][quad of the wave's NTW cout tiles][2]; quad Q of the block = 8 * (NTW * wn + nt) + 2 q + h
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float s = gsum[nt][q], ss = gsq[nt][q];
#pragma unroll
                for (int m = 16; m > 0; m >>= 1) {
                    s += __shfl_xor(s, m);
                    ss += __shfl_xor(ss, m);
                }
                if (col == 0) {
                    int quad = 8 * nt + 2 * q + h;
                    red[(w * (NTW * 8) + quad) * 2 + 0] = s;
                    red[(w * (NTW * 8) + quad) * 2 + 1] = ss;
                }
            }
        __syncthreads();
        if (tid < NT * 16) {
            const int g = tid / (NTW * 16), e = tid % (NTW * 16);   // cout wave group, (local quad, sum | sumsq)
            float t = 0.0f;
#pragma unroll
            for (int m = 0; m < 4 / WN; ++m) t += red[(m * WN + g) * (NTW * 16) + e];   // waves with wn == g, fixed order
            a.partial[((long long)b * a.nblk + blk) * (NT * 16) + tid] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------
// down2_act_split_kernel (round 4): down2_act_kernel for precision = "split".
//
// Tensors hold [hi (C) | lo (C)] fp16 pairs per voxel line (value = hi + lo), the weight comes as all hi fragments
// followed by all lo fragments (sk_conv3d_pack_weight_split_host, ksize 2), a K step is three MFMAs -- w_lo x_hi,
// w_hi x_lo, w_hi x_hi -- exactly as in gather_gemm_kernel<.., SPLIT>, which ran these layers until round 3 behind a
// separate gn_silu_split pass over the skip tensor (read + write of 14.7 GB per 64 production tiles for enc0.1's
// output).  Here the RAW tensor comes in and is activated while it is staged, as in down2_act_kernel:
//   * a stage (one (dx, dy)) holds the two z-adjacent input voxels of NV output voxels as TWO images of 16 KiB -- the hi
//     halves and the lo halves, each with down2_act_kernel's row layout (rows of 4 CIN bytes = 128 | 256, 16-byte chunk
//     c of row n at chunk slot c ^ g(n): conflict-free ds_read_b128 lane groups);
//   * LDS-DMA instruction t of a wave fills slots 64 t .. 64 t + 63 and the lo image lies 16 instructions behind the hi
//     image, so a lane's slot of instruction t + 16 is the lo half of the SAME eight channels of the same voxel as its
//     slot of instruction t: after its own counted vmcnt wait a lane holds both halves of the values it staged and
//     evaluates silu(a (hi + lo) + b) ONCE (gn_silu_split_kernel's arithmetic, op for op), writes hi' = fp16(s) and
//     lo' = fp16(s - hi') back to LDS and to the tensor (the decoder's skip conv reads it activated);
//   * a workgroup runs kSub sub-blocks of NV = 128 | 64 output voxels one after the other, so that it owns the same
//     256 | 128 output voxels -- the same partial-sum rows -- as a down2_act_kernel workgroup (sk_conv3d_num_blocks).
// ------------------------------------------------------------------------------------------
template <int COUT, int CIN>
__global__ void __launch_bounds__(256, 2) down2_act_split_kernel(DownArgs a) {
    constexpr int NT = COUT / 32;
    constexpr int VB = 4 * CIN;                // bytes per voxel line [hi | lo]
    constexpr int RBH = 4 * CIN;               // bytes per row of ONE image: two voxels x CIN fp16 (128 | 256)
    constexpr int CPR = RBH / 16;              // chunks per image row (8 | 16)
    constexpr int CPVH = CIN / 8;              // chunks per voxel and half (4 | 8)
    constexpr int kImage = 16384;              // bytes per image: NV rows
    constexpr int NV = kImage / RBH;           // output voxels per sub-block (128 | 64)
    constexpr int kSub = 2;                    // sub-blocks per workgroup: 256 | 128 output voxels, as down2_act_kernel
    constexpr int WN = NT / 2;                 // wave groups along cout (1 | 2)
    constexpr int NTW = NT / WN;               // cout tiles per wave (2)
    constexpr int PV = (NV / 32) / (4 / WN);   // column tiles per wave (1)
    constexpr int NKS = CIN / 16;              // K steps per input voxel and half
    constexpr int kStage = 2 * kImage;
    constexpr int kOvs = COUT * 4;             // bytes per output voxel line [hi | lo]
    static_assert((CIN == 32 && COUT == 64) || (CIN == 64 && COUT == 128), "the two stride-2 layers of the network");
    static_assert(PV == 1 && NTW == 2, "one column tile, two cout tiles per wave");
    extern __shared__ __attribute__((aligned(16))) char dlds[];   // [2 stages][hi image | lo image] + 4 epilogue pads
    __shared__ float red[4 * 2 * 16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int wn = w % WN, wm = w / WN;
    const int b = blockIdx.x / a.nblk, blk = blockIdx.x % a.nblk;
    const int Yi = 2 * a.Yo, Zi = 2 * a.Zo;
    const long long nvox = (long long)a.Xo * a.Yo * a.Zo;
    char* inb = a.in + (long long)b * 8 * nvox * VB;
    char* outb = a.out + (long long)b * nvox * kOvs;
    const bool raw = a.affine != nullptr;
    const char* wlo = a.wpk + (long long)(8 * NKS) * NT * 1024;   // the lo fragments follow the 8 taps x NKS hi steps
    // swizzle term of image row n (down2_act_kernel's): 128-byte rows alternate bank halves, 256-byte rows start at bank 0
    auto gsw = [](int n) { return CPR == 8 ? ((n >> 1) & 7) : (n & 15); };

    // this thread's slots of an image: slot = tid + 256 j (j < 4) -> row n_j = tid / CPR + (256 / CPR) j, chunk slot tid % CPR;
    // the swizzle term is the same for every j, hence one source chunk: voxel dz = csrc / CPVH of the row, channel octet csrc % CPVH
    const int csrc = (tid % CPR) ^ gsw(tid / CPR);
    const int src_off = (csrc / CPVH) * VB + (csrc % CPVH) * 16;   // byte offset of the hi chunk in the two-voxel row; lo: + VB / 2
    float ga[8], gb[8];
    if (raw) {
        const int c0 = (csrc % CPVH) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ga[j] = a.affine[(long long)b * 2 * CIN + c0 + j];
            gb[j] = a.affine[(long long)b * 2 * CIN + CIN + c0 + j];
        }
    }
    float gsum[NTW][4], gsq[NTW][4];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) gsum[nt][q] = gsq[nt][q] = 0.0f;
    char* pad = dlds + 2 * kStage + w * kPadBytes;
    const int rv = lane >> 2, rc = lane & 3;

    for (int sub = 0; sub < kSub; ++sub) {
        const long long v0 = ((long long)blk * kSub + sub) * NV;
        if (v0 >= nvox) break;   // block-uniform
        long long vin[4];        // input voxel index of tap (0, 0, 0) of row n_j, or -1 beyond the tensor
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long v = v0 + tid / CPR + (256 / CPR) * j;
            if (v < nvox) {
                const int zo = (int)(v % a.Zo);
                const long long t = v / a.Zo;
                const int yo = (int)(t % a.Yo), xo = (int)(t / a.Yo);
                vin[j] = ((long long)(2 * xo) * Yi + 2 * yo) * Zi + 2 * zo;
            } else {
                vin[j] = -1;
            }
        }
        auto issue_stage = [&](int st) {
            const long long toff = ((long long)(st >> 1) * Yi + (st & 1)) * Zi;   // (dx, dy)
            char* lbase = dlds + (st & 1) * kStage;
#pragma unroll
            for (int part = 0; part < 2; ++part)      // the hi image, then the lo image 16 instructions behind it
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const char* g = vin[j] >= 0 ? inb + (vin[j] + toff) * VB + src_off + part * (VB / 2) : a.zeros + lane * 16;
                    dma16(g, lbase + part * kImage + (w + 4 * j) * 1024);
                }
        };
        f32x16 acc[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + 32 * (NTW * wn + nt) + 8 * q + 4 * h);
                acc[nt][4 * q] = bv[0];
                acc[nt][4 * q + 1] = bv[1];
                acc[nt][4 * q + 2] = bv[2];
                acc[nt][4 * q + 3] = bv[3];
            }
        // weight fragments one K step ahead: [buffer][cout tile][hi | lo]
        half8 wf[2][NTW][2];
        auto wload = [&](int s, half8 (&dst)[NTW][2]) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const long long off = ((long long)s * NT + NTW * wn + nt) * 1024 + lane * 16;
                dst[nt][0] = *reinterpret_cast<const half8*>(a.wpk + off);
                dst[nt][1] = *reinterpret_cast<const half8*>(wlo + off);
            }
        };
        __syncthreads();   // the previous sub-block's epilogue pads / stage buffers are free
        issue_stage(0);
        for (int st = 0; st < 4; ++st) {
            if (st + 1 < 4) {
                issue_stage(st + 1);   // its buffer was last read by stage st - 1's MFMAs: the closing barrier below
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // everything older than those 8 pieces has landed
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            wload(st * 2 * NKS, wf[0]);
            char* buf = dlds + (st & 1) * kStage;
            if (raw) {
                const long long toff = ((long long)(st >> 1) * Yi + (st & 1)) * Zi;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    half8* lph = reinterpret_cast<half8*>(buf + (tid + 256 * j) * 16);
                    half8* lpl = reinterpret_cast<half8*>(buf + kImage + (tid + 256 * j) * 16);
                    const half8 vh = *lph, vl = *lpl;
                    half8 rh, rl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {   // gn_silu_split_kernel's arithmetic, op for op
                        const float y = fmaf(ga[e], (float)vh[e] + (float)vl[e], gb[e]);
                        const float sv = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
                        rh[e] = sk::round_t16(sv);
                        rl[e] = (t16)(sv - (float)rh[e]);
                    }
                    *lph = rh;
                    *lpl = rl;
                    if (a.writeback == 1 && vin[j] >= 0) {
                        char* g = inb + (vin[j] + toff) * VB + src_off;
                        *reinterpret_cast<half8*>(g) = rh;
                        *reinterpret_cast<half8*>(g + VB / 2) = rl;
                    }
                    if (a.writeback == 2 && vin[j] >= 0) {
                        // precision "mix8": the activated line as [hi (CIN) | per 32-channel chunk: x8 = e4m3(16 x) | lo8 = e4m3(2^15 (x - hi))]
                        // (sk_groupnorm_silu_mix8's store).  The 8-bit pieces land on raw lo halves that OTHER lanes staged: the
                        // CPR lanes that stage a row's two voxel lines sit in one wave, whose LDS-DMA of this stage has landed
                        // (the vmcnt wait above) before any of its lanes gets here.
                        float xs[8], ls[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float y = fmaf(ga[e], (float)vh[e] + (float)vl[e], gb[e]);
                            const float sv = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
                            xs[e] = fminf(fmaxf(sv * 16.0f, -448.0f), 448.0f);
                            ls[e] = fminf(fmaxf((sv - (float)rh[e]) * 32768.0f, -448.0f), 448.0f);
                        }
                        int px[2] = {0, 0}, pl[2] = {0, 0};
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2) {
                            px[k2] = __builtin_amdgcn_cvt_pk_fp8_f32(xs[4 * k2], xs[4 * k2 + 1], px[k2], false);
                            px[k2] = __builtin_amdgcn_cvt_pk_fp8_f32(xs[4 * k2 + 2], xs[4 * k2 + 3], px[k2], true);
                            pl[k2] = __builtin_amdgcn_cvt_pk_fp8_f32(ls[4 * k2], ls[4 * k2 + 1], pl[k2], false);
                            pl[k2] = __builtin_amdgcn_cvt_pk_fp8_f32(ls[4 * k2 + 2], ls[4 * k2 + 3], pl[k2], true);
                        }
                        const int oct = csrc % CPVH;
                        char* line = inb + (vin[j] + toff + csrc / CPVH) * VB;
                        *reinterpret_cast<half8*>(line + 16 * oct) = rh;
                        char* part8 = line + VB / 2 + 64 * (oct >> 2) + 8 * (oct & 3);
                        *reinterpret_cast<int2*>(part8) = make_int2(px[0], px[1]);
                        *reinterpret_cast<int2*>(part8 + 32) = make_int2(pl[0], pl[1]);
                    }
                }
            }
            __syncthreads();
            // K steps of the stage: (dz, ks); weight step index = st * 2 NKS + dz * NKS + ks
            const int n = 32 * wm + col;
            const char* rowp = buf + n * RBH;
            const int sw = gsw(n);
#pragma unroll
            for (int k = 0; k < 2 * NKS; ++k) {
                if (k + 1 < 2 * NKS) wload(st * 2 * NKS + k + 1, wf[(k + 1) & 1]);
                const int coff = ((2 * k + h) ^ sw) * 16;   // chunk (dz, ks, h) of the row = 2 k + h, as in down2_act_kernel
                const half8 bh = *reinterpret_cast<const half8*>(rowp + coff);
                const half8 bl = *reinterpret_cast<const half8*>(rowp + kImage + coff);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    acc[nt] = SK_MFMA_32x32x16_T16(wf[k & 1][nt][1], bh, acc[nt], 0, 0, 0);
                    acc[nt] = SK_MFMA_32x32x16_T16(wf[k & 1][nt][0], bl, acc[nt], 0, 0, 0);
                    acc[nt] = SK_MFMA_32x32x16_T16(wf[k & 1][nt][0], bh, acc[nt], 0, 0, 0);
                }
            }
            __syncthreads();   // stage buffer free for the DMA of stage st + 2
        }

        // ---- epilogue: hi then lo halves through the per-wave pad, 16-byte stores; statistics of the fp32 accumulators
        const long long vbase = v0 + (long long)wm * 32;
        const bool okp = vbase + col < nvox;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
            for (int part = 0; part < 2; ++part) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x0 = acc[nt][4 * q], x1 = acc[nt][4 * q + 1];
                    const float x2 = acc[nt][4 * q + 2], x3 = acc[nt][4 * q + 3];
                    half4 hv = {(t16)x0, (t16)x1, (t16)x2, (t16)x3};
                    if (part == 1)
                        hv = half4{(t16)(x0 - (float)hv[0]), (t16)(x1 - (float)hv[1]), (t16)(x2 - (float)hv[2]),
                                   (t16)(x3 - (float)hv[3])};
                    *reinterpret_cast<half4*>(pad + col * kPadStride + ((q ^ ((col >> 1) & 3)) * 16) + 8 * h) = hv;
                    if (part == 0 && okp) {
                        gsum[nt][q] += (x0 + x1) + (x2 + x3);
                        gsq[nt][q] += (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
                    }
                }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int vv = rv + 16 * hh;
                    const half8 line = *reinterpret_cast<const half8*>(pad + vv * kPadStride + ((rc ^ ((vv >> 1) & 3)) * 16));
                    const long long v = vbase + vv;
                    if (v < nvox) *reinterpret_cast<half8*>(outb + v * kOvs + part * (COUT * 2) + (NTW * wn + nt) * 64 + rc * 16) = line;
                }
            }
        }
    }
    if (a.partial) {
        // red[wave][quad of the wave's NTW cout tiles][2]; quad Q of the block = 8 * (NTW * wn + nt) + 2 q + h
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float s = gsum[nt][q], ss = gsq[nt][q];
#pragma unroll
                for (int m = 16; m > 0; m >>= 1) {
                    s += __shfl_xor(s, m);
                    ss += __shfl_xor(ss, m);
                }
                if (col == 0) {
                    const int quad = 8 * nt + 2 * q + h;
                    red[(w * (NTW * 8) + quad) * 2 + 0] = s;
                    red[(w * (NTW * 8) + quad) * 2 + 1] = ss;
                }
            }
        __syncthreads();
        if (tid < NT * 16) {
            const int g = tid / (NTW * 16), e = tid % (NTW * 16);   // cout wave group, (local quad, sum | sumsq)
            float t = 0.0f;
#pragma unroll
            for (int m = 0; m < 4 / WN; ++m) t += red[(m * WN + g) * (NTW * 16) + e];   // waves with wn == g, fixed order
            a.partial[((long long)b * a.nblk + blk) * (NT * 16) + tid] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------
// host-side planning
// ------------------------------------------------------------------------------------------
struct Plan {
    int mode, TZ, nzc, pitch, nposp, npatch, XC, nxc, xs;
    size_t lds;
};

int conv3_xs(int cout) { return cout == 128 ? 2 : 4; }  // keeps P*XS*16 accumulators <= 128

int make_plan(Plan& p, int Xt, int Yt, int Zt, int cout, int B) {
    p.xs = conv3_xs(cout);
    if (Zt <= 40) {
        p.mode = 0;
        p.TZ = Zt;
        p.nzc = 1;
        p.pitch = Zt;  // no z halo: see the kernel's linear-mode comment
        long long nv = (long long)Yt * Zt;
        p.npatch = (int)((nv + kPatch - 1) / kPatch);
        p.nposp = (kPatch + 2 * Zt + 2 + 15) / 16 * 16;  // 128 voxels + one row and one voxel on each side
    } else {
        // Rectangle patches TY x TZ with a one-voxel halo.  COUT 32 (conv3_m16_kernel, round 4): 8 x 16 -- a 16-column MFMA
        // operand is one z segment of one row, the region is 10 x 18 = 180 positions instead of 6 x 34 = 204, and a
        // six-plane ring (XS = 4) fits two workgroups per CU (round 1-3: 4 x 32 rectangles, 208 positions, XS = 3 at
        // +13-20 % time).  COUT 64 / 128 (conv3_kernel, 32-column operands): 4 x 32.
        p.mode = 1;
        p.TZ = cout == 32 ? 16 : 32;
#ifdef SK_TUNING
        if (getenv("SK_CONV_RECT_4X32")) p.TZ = 32;   // A/B: rounds 1-3's rectangles for COUT 32 as well
#endif
        p.nzc = (Zt + p.TZ - 1) / p.TZ;
        p.pitch = p.TZ + 2;
        int TY = kPatch / p.TZ;
        p.npatch = ((Yt + TY - 1) / TY) * p.nzc;
        p.nposp = ((TY + 2) * p.pitch + 15) / 16 * 16;
    }
    if (p.nposp > 64 * kMaxDma) return -1;
    // the 16x16x32 kernels (COUT 32) transpose their results with v_permlane16_swap: no LDS pads
    const size_t pads = cout == 32 ? 0 : 4 * kPadBytes;
    p.lds = (size_t)(p.xs + 2) * (p.nposp + sk::kZeroPos) * kPosBytes + pads;
    // Two workgroups per CU need <= 80 KiB each.  The rectangle patches (208 positions a plane) miss that with a
    // 6-plane ring: run them with XS = 3 (5 planes); measured on the 512x512x128 tile (SK_CONV_RECT_XS4=1 to compare).
    bool keep_xs4 = false;
#ifdef SK_TUNING
    keep_xs4 = getenv("SK_CONV_RECT_XS4") != nullptr;
#endif
    if (p.xs == 4 && p.lds > 80 * 1024 && !keep_xs4) {
        p.xs = 3;
        p.lds = (size_t)(p.xs + 2) * (p.nposp + sk::kZeroPos) * kPosBytes + pads;
    }
    // x-chunks: enough workgroups to fill the CUs (2 per CU) several times over at the batch sizes the pipeline runs
    // (8-32 tiles).  The cut is a function of the tile geometry ONLY, not of B: the GroupNorm partial sums are fp32 sums
    // per workgroup, so a batch-dependent cut would make a tile's statistics -- and through them its output bits --
    // depend on how many tiles share its launch (tests/test_hip_geometry.py: batch invariance).
    (void)B;
    const int kPlanBatch = 8;
    int target = 256 * 2 * 8;   // (x 6 left the 64-channel layers of a 256^3 training crop, batch 1, with 384 workgroups for 512 slots)
    int nxc = (target + p.npatch * kPlanBatch - 1) / (p.npatch * kPlanBatch);
    // COUT 32 on linear patches (the full-resolution layers of the production tile): half as many, twice as long chunks.
    // conv3_px_kernel pays a fixed price per workgroup (weights into registers / LDS, four planes staged, the planes at
    // the chunk ends): x-chunks of 76 instead of 36 planes measured -2 % (enc0.1) / -4 % (dec0.1) per 64 tiles, and two
    // chunks of 150 against five of 60 another -0.5 % / -1.5 % (one of 300: -1 % / -2.7 %, but a batch of 8 or 16 tiles
    // then leaves CUs idle: 376 / 752 workgroups for 512 slots)
    if (cout == 32 && p.mode == 0) nxc = nxc >= 4 ? 2 : (nxc + 1) / 2;
    int max_nxc = (Xt + 2 * p.xs - 1) / (2 * p.xs);  // at least two steps per chunk
    if (nxc > max_nxc) nxc = max_nxc;
    if (nxc < 1) nxc = 1;
    int XC = (Xt + nxc - 1) / nxc;
#ifdef SK_TUNING
    if (const char* e = getenv("SK_CONV_XC")) {  // tuning experiments
        const int v = atoi(e);
        if (v >= p.xs && v <= Xt + p.xs) XC = v;
    }
#endif
    XC = (XC + p.xs - 1) / p.xs * p.xs;
    p.XC = XC;
    p.nxc = (Xt + XC - 1) / XC;
    return 0;
}

// mix8 weight image (sk_conv3d_pack_weight_mix8_host): the fp16 fragments of w_hi (54 KiB per 32 x 32 channels), then fp8
// fragments of 2 KiB -- 32 -> 32 (conv3_m16_kernel, K = 128 = two tap rows): 5 row pairs x 2 cout halves x 3 x taps;
// wider (conv3_kernel, K = 64 = one tap): per 32-channel chunk 27 taps x cout / 32 tiles
constexpr int kMix8Fp16Bytes = 54 * 1024, kMix8Fp8Bytes = 5 * 2 * 3 * 2048;
inline int mix8_fp16_bytes(int cout, int cin) { return (cout / 32) * (cin / 32) * kMix8Fp16Bytes; }
inline int mix8_fp8_bytes(int cout, int cin) { return cout == 32 ? kMix8Fp8Bytes : (cin / 32) * 27 * (cout / 32) * 2048; }

template <int XS, int RES = 0, bool SPLIT = false, int WL = 0, bool MIX8 = false>
int launch_conv3_m16(const Conv3Args& a, const Plan& p, hipStream_t stream) {
    auto kern = conv3_m16_kernel<32, XS, RES, SPLIT, WL, MIX8>;
    const size_t lds = p.lds + (RES > 0 ? WL * 6144 : 0);   // the ring + the tap rows kept in LDS behind it
    if (lds > 48 * 1024)
        SK_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds));
    unsigned grid = (unsigned)(p.npatch * p.nxc * a.B);
    kern<<<grid, 256, lds, stream>>>(a);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

// conv3_px_kernel is built for the plane geometry of the production tile: linear mode with 176 positions per plane
// (Zt 16 .. 20: 128 + 2 Zt + 2 positions plus six padding positions -- bias, GroupNorm coefficients, the 256-byte zero window).  The first
// kPxResidentHalfRows half tap rows (3 fragments each) in registers, the other 18 - that in LDS: 4 plane slots (44 KiB) +
// 12 half rows (36 KiB) = 80 KiB, two workgroups per CU.  (7 resident half rows, 77 KiB: 18 spilled registers, two scratch
// reloads per step: enc0.1 5.15 -> 5.92 ms per 64 tiles.)
#ifndef SK_PX_RESH
#define SK_PX_RESH 6
#endif
constexpr int kPxResidentHalfRows = SK_PX_RESH;   // (tools/ A/B: -DSK_PX_RESH=6 is the 80 KiB form)
constexpr int kPxPositions = 176;
constexpr size_t kPxLds = (size_t)4 * kPxPositions * kPosBytes + (size_t)(18 - kPxResidentHalfRows) * 3072;
// padding positions behind the `needed` ones: 2 (bias / GroupNorm coefficients) + the 4-position zero window
bool conv3_px_covers(const Plan& p, int Zt, bool /*raw*/) { return p.mode == 0 && p.nposp == kPxPositions && kPatch + 2 * Zt + 2 + 6 <= kPxPositions; }

int launch_conv3_px(const Conv3Args& a, const Plan& p, hipStream_t stream) {
    auto kern = conv3_px_kernel<kPxResidentHalfRows, kPxPositions>;
    const size_t lds = kPxLds;
    static_assert(kPxLds <= 80 * 1024, "two workgroups per CU");
    SK_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned grid = (unsigned)(p.npatch * p.nxc * a.B);
    kern<<<grid, 256, lds, stream>>>(a);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int launch_conv3_pxm(const Conv3Args& a, const Plan& p, hipStream_t stream) {
    auto kern = conv3_pxm_kernel<kPxResidentHalfRows, kPxPositions>;
    SK_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPxLds));
    unsigned grid = (unsigned)(p.npatch * p.nxc * a.B);
    kern<<<grid, 256, kPxLds, stream>>>(a);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

template <int COUT, int XS, int RES = 0, bool SPLIT = false, bool MIX8 = false>
int launch_conv3(const Conv3Args& a, const Plan& p, hipStream_t stream) {
    auto kern = conv3_kernel<COUT, XS, RES, SPLIT, MIX8>;
    if (p.lds > 48 * 1024)
        SK_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)p.lds));
    unsigned grid = (unsigned)(p.npatch * p.nxc * a.B);
    kern<<<grid, 256, p.lds, stream>>>(a);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

}  // namespace

extern "C" {

int sk_conv3d_num_blocks(int B, int ox, int oy, int oz, int cout, int ksize) {
    if (ksize == 3) {
        Plan p;
        if (make_plan(p, ox, oy, oz, cout, B)) return -1;
        return p.npatch * p.nxc;
    }
    int pv = cout == 128 ? 1 : 2;
    long long nv = (long long)ox * oy * oz;
    return (int)((nv + 128 * pv - 1) / (128 * pv));
}

int64_t sk_conv3d_pack_weight_host(const float* w, int cout, int cin, int ksize, void* dst) {
    // torch layout (cout, cin, kx, ky, kz) fp32 -> MFMA A fragments (1 KiB = 64 lanes x 8 halves).
    // ksize 1/2 (v_mfma_f32_32x32x16_f16): lane l holds W[cout = 32nt + (l&31)][cin = c0 + 8(l>>5) + j],
    // order [tap][ks(cin/16)][nt]; ksize 3: see emit16 below
    if (!(ksize == 1 || ksize == 2 || ksize == 3) || cout % 32 || cin % (ksize == 3 ? 32 : 16)) {
        sk::set_error("sk_conv3d_pack_weight_host: unsupported shape cout=%d cin=%d k=%d", cout, cin, ksize);
        return SK_ERR_ARG;
    }
    const int NT = cout / 32, k3 = ksize * ksize * ksize;
    int64_t nfrag = (int64_t)k3 * (cin / 16) * NT;
    if (!dst) return nfrag * 1024;
    t16* out = (t16*)dst;
    auto W = [&](int co, int ci, int kx, int ky, int kz) {
        return w[((((int64_t)co * cin + ci) * ksize + kx) * ksize + ky) * ksize + kz];
    };
    int64_t f = 0;
    auto emit = [&](int nt, int c0, int kx, int ky, int kz) {
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j)
                out[f * 512 + l * 8 + j] = (t16)(W(32 * nt + (l & 31), c0 + 8 * (l >> 5) + j, kx, ky, kz));
        ++f;
    };
    // ksize 3 with cout 32 runs on v_mfma_f32_16x16x32_f16: fragment = [16 cout][32 channels of the chunk], lane l
    // holds W[cout = 16i + (l&15)][cin = 32ch + 8(l>>4) + j]; order [chunk32][dy*3+dz][i(2)][dx]
    auto emit16 = [&](int nt, int i, int c0, int kx, int ky, int kz) {
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j)
                out[f * 512 + l * 8 + j] = (t16)(W(32 * nt + 16 * i + (l & 15), c0 + 8 * (l >> 4) + j, kx, ky, kz));
        ++f;
    };
    if (ksize == 3 && cout == 32) {   // the COUT-32 conv kernel is the 16x16x32 one
        for (int ch = 0; ch < cin / 32; ++ch)
            for (int dydz = 0; dydz < 9; ++dydz)
                for (int i = 0; i < 2; ++i)
                    for (int dx = 0; dx < 3; ++dx)
                        for (int nt = 0; nt < NT; ++nt) emit16(nt, i, ch * 32, dx, dydz / 3, dydz % 3);
    } else if (ksize == 3) {          // 32x32x16: order [chunk32][dy*3+dz][ks(2)][dx][nt]
        for (int ch = 0; ch < cin / 32; ++ch)
            for (int dydz = 0; dydz < 9; ++dydz)
                for (int ks = 0; ks < 2; ++ks)
                    for (int dx = 0; dx < 3; ++dx)
                        for (int nt = 0; nt < NT; ++nt) emit(nt, ch * 32 + ks * 16, dx, dydz / 3, dydz % 3);
    } else {
        for (int tap = 0; tap < k3; ++tap)
            for (int ks = 0; ks < cin / 16; ++ks)
                for (int nt = 0; nt < NT; ++nt)
                    emit(nt, ks * 16, tap / (ksize * ksize), (tap / ksize) % ksize, tap % ksize);
    }
    return nfrag * 1024;
}

static int launch_down2(const void* in, const float* affine, int writeback, const void* weight, const float* bias, void* out,
                        int B, int ox, int oy, int oz, int cin, int cout, float* gn_partial, void* zeros, hipStream_t stream,
                        bool split = false) {
    SK_CHECK_ARG(in && weight && bias && out && zeros, "down conv: NULL pointer (the zero page is required)");
    SK_CHECK_ARG((cin == 32 && cout == 64) || (cin == 64 && cout == 128), "down conv: (cin, cout) must be (32, 64) or (64, 128)");
    SK_CHECK_ARG(!writeback || affine, "down conv: write-back needs the affine of the raw input");
    DownArgs a{};
    a.in = (char*)in;
    a.affine = affine;
    a.wpk = (const char*)weight;
    a.bias = bias;
    a.out = (char*)out;
    a.partial = gn_partial;
    a.zeros = (const char*)zeros;
    a.B = B;
    a.Xo = ox;
    a.Yo = oy;
    a.Zo = oz;
    a.writeback = writeback;
    a.nblk = sk_conv3d_num_blocks(B, ox, oy, oz, cout, 2);   // 256 (cout 64) / 128 (cout 128) voxels per block: NV
    const int lds = 2 * 32768 + 4 * kPadBytes;
    const unsigned grid = (unsigned)(a.nblk * B);
    if (split) {
        if (cin == 32) {
            auto kern = down2_act_split_kernel<64, 32>;
            SK_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            kern<<<grid, 256, lds, stream>>>(a);
        } else {
            auto kern = down2_act_split_kernel<128, 64>;
            SK_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            kern<<<grid, 256, lds, stream>>>(a);
        }
    } else if (cin == 32) {
        auto kern = down2_act_kernel<64, 32>;
        SK_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        kern<<<grid, 256, lds, stream>>>(a);
    } else {
        auto kern = down2_act_kernel<128, 64>;
        SK_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        kern<<<grid, 256, lds, stream>>>(a);
    }
    SK_CHECK_LAUNCH();
    return SK_OK;
}

static int conv3d_impl(const sk_conv_src* srcs, int n_src, const void* weight, const float* bias, void* out,
                       int B, int ox, int oy, int oz, int cout, int ksize, float* gn_partial,
                       void* zeros, void* stream_, const bool split, const int* store_box = nullptr,
                       const bool mix8 = false, const int w8_scale_exp = 0) {
    hipStream_t stream = (hipStream_t)stream_;
    const int lanes = split ? 2 : 1;   // fp16 values per logical channel in a voxel line: [hi | lo]
    SK_CHECK_ARG(srcs && weight && bias && out, "sk_conv3d: NULL pointer");
    SK_CHECK_ARG(n_src == 1 || n_src == 2, "sk_conv3d: n_src must be 1 or 2");
    SK_CHECK_ARG(B > 0 && ox > 0 && oy > 0 && oz > 0, "sk_conv3d: bad output extents");
    SK_CHECK_ARG(cout == 32 || cout == 64 || cout == 128, "sk_conv3d: cout must be 32, 64 or 128");
    if (ksize == 3) {
        SK_CHECK_ARG(zeros, "sk_conv3d: zero page is NULL");
        Conv3Args a{};
        int cin = 0;
        for (int i = 0; i < n_src; ++i) {
            SK_CHECK_ARG(srcs[i].data && srcs[i].c > 0 && srcs[i].c % kChunk == 0,
                         "sk_conv3d: source %d must have a multiple of 32 channels", i);
            SK_CHECK_ARG(srcs[i].affine == nullptr || !split,
                         "sk_conv3d_split: sources must be activated (affine must be NULL)");
            SK_CHECK_ARG(srcs[i].affine == nullptr || cout == 32,
                         "sk_conv3d: ksize 3 activates a RAW source in LDS only in the COUT 32 kernel (for wider layers the "
                         "separate pass is cheaper: activate the tensor with sk_groupnorm_silu first)");
            a.act[i] = srcs[i].affine;   // RAW source: activated in LDS by the lanes that stage it
            SK_CHECK_ARG(i == 1 || !srcs[i].upsample, "sk_conv3d: only the second source may be upsampled");
            int up = srcs[i].upsample ? 1 : 0;
            SK_CHECK_ARG(!up || (ox % 2 == 0 && oy % 2 == 0 && oz % 2 == 0),
                         "sk_conv3d: upsampled source needs even output extents");
            int xs = up ? ox / 2 : ox, ys = up ? oy / 2 : oy, zs = up ? oz / 2 : oz;
            a.src[i].data = (const char*)srcs[i].data;
            a.src[i].C = srcs[i].c * lanes;     // halves per voxel line
            a.src[i].up = up;
            a.src[i].Ys = ys;
            a.src[i].Zs = zs;
            a.src[i].plane = (long long)ys * zs * srcs[i].c * lanes * 2;
            a.src[i].batch = a.src[i].plane * xs;
            cin += srcs[i].c;
        }
        if (n_src == 1) {
            a.src[1] = a.src[0];
            a.act[1] = a.act[0];
        }
        a.c0chunks = srcs[0].c / kChunk;
        // phase chunks of a step.  Plain: the 32-channel chunks of the concatenated sources in order.  Split: per
        // logical chunk three phases -- (x_hi, w_lo), (x_hi again: no DMA, w_hi), (x_lo, w_hi); the packed weight
        // (sk_conv3d_pack_weight_split_host) holds the matching [lo, hi, hi] fragment sets in this order.
        a.nchunks = 0;
        for (int k = 0; k < cin / kChunk; ++k) {
            const int si = k < a.c0chunks ? 0 : 1;
            const unsigned off = (unsigned)(k - (si ? a.c0chunks : 0)) * kChunk * 2;
            const unsigned lo_off = off + (unsigned)srcs[si].c * 2;
            SK_CHECK_ARG(a.nchunks + (split ? 3 : 1) <= kMaxChunks, "sk_conv3d: too many input channels (%d)", cin);
            if (!split) {
                a.chinfo[a.nchunks++] = (unsigned)si | (off << 8);
            } else if (mix8) {   // [hi fp16 | x8 | lo8] lines: the fp16 phase, then ONE fp8 phase over the 64 bytes behind the hi halves
                a.chinfo[a.nchunks++] = (unsigned)si | (off << 8);
                a.chinfo[a.nchunks++] = (unsigned)si | 4u | (lo_off << 8);
            } else {
                a.chinfo[a.nchunks++] = (unsigned)si | (off << 8);
                a.chinfo[a.nchunks++] = (unsigned)si | 2u | (off << 8);
                a.chinfo[a.nchunks++] = (unsigned)si | (lo_off << 8);
            }
        }
        a.wpk = (const char*)weight;
        a.bias = bias;
        a.out = (char*)out;
        a.partial = gn_partial;
        a.zeros = (const char*)zeros;
        a.B = B;
        a.Xt = ox;
        a.Yt = oy;
        a.Zt = oz;
        Plan p;
        SK_CHECK_ARG(make_plan(p, ox, oy, oz, cout, B) == 0, "sk_conv3d: plane geometry (%d,%d) unsupported", oy, oz);
        a.XC = p.XC;
        a.nxc = p.nxc;
        a.npatch = p.npatch;
        a.mode = p.mode;
        a.TZ = p.TZ;
        a.nzc = p.nzc;
        a.pitch = p.pitch;
        a.nposp = p.nposp;
        a.jstep = (p.mode == 1 && p.TZ == 16) ? p.pitch : 16;
        // two-chunk layers only: with four chunks the reuse is 1/12 of the loads and measured +1.6 % time (COUT 128);
        // split mode: the chunk triples carry their own reuse flags
        a.alt = (!split && a.nchunks == 2) ? 1 : 0;
        a.dbg = nullptr;
#ifdef SK_TIMING
        a.dbg = sk::timing_buffer();
        if (const char* e = getenv("SK_CONV_DBG_OFF")) a.dbg_off = atoi(e);
#endif
        a.ablate = 0;
#ifdef SK_TUNING
        if (const char* e = getenv("SK_CONV_ABLATE")) a.ablate = atoi(e);
#endif
        a.has_box = 0;
        if (store_box && cout == 32) {   // honoured by the COUT-32 kernels (conv3_px_kernel, conv3_m16_kernel); wider layers store everything
            for (int k = 0; k < 3; ++k) {
                a.box_lo[k] = store_box[k];
                a.box_hi[k] = store_box[3 + k];
                SK_CHECK_ARG(a.box_lo[k] >= 0 && a.box_lo[k] <= a.box_hi[k], "sk_conv3d_box: bad box on axis %d", k);
            }
            a.has_box = 1;
        }
        if (mix8) {
            SK_CHECK_ARG(split && cin == cout && n_src == 1 && !srcs[0].upsample,
                         "sk_conv3d_mix8: one plain source with as many channels as the output (32 | 64 | 128)");
            SK_CHECK_ARG(w8_scale_exp >= 0 && w8_scale_exp < 64, "sk_conv3d_mix8: bad weight scale exponent %d", w8_scale_exp);
            a.w8_off = mix8_fp16_bytes(cout, cin);
            a.wpk_bytes = a.w8_off + mix8_fp8_bytes(cout, cin);
            a.w8_scale = 0x01010101 * (127 - w8_scale_exp);
            if (cout == 32) {
                bool use_pxm = conv3_px_covers(p, oz, false);
#ifdef SK_TUNING
                if (getenv("SK_CONV_NO_PX")) use_pxm = false;   // A/B: conv3_m16_kernel's mix8 form
#endif
                if (use_pxm) return launch_conv3_pxm(a, p, stream);
                return p.xs == 3 ? launch_conv3_m16<3, 0, true, 0, true>(a, p, stream) : launch_conv3_m16<4, 0, true, 0, true>(a, p, stream);
            }
            if (cout == 64) return p.xs == 3 ? launch_conv3<64, 3, 0, true, true>(a, p, stream) : launch_conv3<64, 4, 0, true, true>(a, p, stream);
            return launch_conv3<128, 2, 0, true, true>(a, p, stream);
        }
        if (split) {
            if (cout == 32) return p.xs == 3 ? launch_conv3_m16<3, 0, true>(a, p, stream) : launch_conv3_m16<4, 0, true>(a, p, stream);
            if (cout == 64) return p.xs == 3 ? launch_conv3<64, 3, 0, true>(a, p, stream) : launch_conv3<64, 4, 0, true>(a, p, stream);
            return launch_conv3<128, 2, 0, true>(a, p, stream);
        }
        bool use_px = cout == 32 && a.nchunks == 1 && n_src == 1 && !srcs[0].upsample && conv3_px_covers(p, oz, a.act[0] != nullptr);
#ifdef SK_TUNING
        if (getenv("SK_CONV_NO_PX")) use_px = false;   // A/B: the single-chunk COUT-32 layers on conv3_m16_kernel
#endif
        if (use_px) return launch_conv3_px(a, p, stream);
        if (cout == 32) {   // 16x16x32 kernel
            if (p.xs == 3) return launch_conv3_m16<3>(a, p, stream);
            if (a.nchunks == 1 && !a.ablate) {   // single chunk: 3 tap rows in registers, 2 more in LDS where they fit
                if (p.lds + 2 * 6144 <= 80 * 1024) return launch_conv3_m16<4, 3, false, 2>(a, p, stream);
                if (p.lds + 1 * 6144 <= 80 * 1024) return launch_conv3_m16<4, 3, false, 1>(a, p, stream);   // the 8 x 16 rectangles
                return launch_conv3_m16<4, 3>(a, p, stream);
            }
            return launch_conv3_m16<4>(a, p, stream);
        }
        if (cout == 64) return p.xs == 3 ? launch_conv3<64, 3>(a, p, stream) : launch_conv3<64, 4>(a, p, stream);
        return launch_conv3<128, 2>(a, p, stream);
    }
    SK_CHECK_ARG(ksize == 1 || ksize == 2, "sk_conv3d: ksize must be 1, 2 or 3");
    SK_CHECK_ARG(n_src == 1 && !srcs[0].upsample, "sk_conv3d: ksize %d takes one plain source", ksize);
    SK_CHECK_ARG(srcs[0].c % 16 == 0, "sk_conv3d: cin must be a multiple of 16");
    if (!split && ksize == 2 && ((srcs[0].c == 32 && cout == 64) || (srcs[0].c == 64 && cout == 128)))
        return launch_down2(srcs[0].data, srcs[0].affine, 0, weight, bias, out, B, ox, oy, oz, srcs[0].c, cout, gn_partial,
                            zeros, stream);
    GatherArgs g{};
    g.in = (const char*)srcs[0].data;
    g.affine = srcs[0].affine;
    g.wpk = (const char*)weight;
    g.bias = bias;
    g.out = (char*)out;
    g.partial = gn_partial;
    g.B = B;
    g.Xo = ox;
    g.Yo = oy;
    g.Zo = oz;
    g.Xi = ox * ksize;
    g.Yi = oy * ksize;
    g.Zi = oz * ksize;
    g.Cin = srcs[0].c;
    g.ksize = ksize;
    g.nblk = sk_conv3d_num_blocks(B, ox, oy, oz, cout, ksize);
    unsigned grid = (unsigned)(g.nblk * B);
    const int nsteps = ksize * ksize * ksize * (g.Cin / 16);
    SK_CHECK_ARG(g.Cin <= 128 && nsteps % 2 == 0, "sk_conv3d: ksize %d needs 32 <= cin <= 128 (got %d)", ksize, g.Cin);
    const bool deep = nsteps % 4 == 0;  // pipeline depth 4 (2 only for the 32-channel pointwise case)
    if (split) {   // depth 2: twice the fragments in flight per step
        if (cout == 32)
            gather_gemm_kernel<32, 2, 2, true><<<grid, 256, 0, stream>>>(g);
        else if (cout == 64)
            gather_gemm_kernel<64, 2, 2, true><<<grid, 256, 0, stream>>>(g);
        else
            gather_gemm_kernel<128, 1, 2, true><<<grid, 256, 0, stream>>>(g);
        SK_CHECK_LAUNCH();
        return SK_OK;
    }
    if (cout == 32)
        deep ? gather_gemm_kernel<32, 2, 4><<<grid, 256, 0, stream>>>(g) : gather_gemm_kernel<32, 2, 2><<<grid, 256, 0, stream>>>(g);
    else if (cout == 64)
        deep ? gather_gemm_kernel<64, 2, 4><<<grid, 256, 0, stream>>>(g) : gather_gemm_kernel<64, 2, 2><<<grid, 256, 0, stream>>>(g);
    else
        deep ? gather_gemm_kernel<128, 1, 4><<<grid, 256, 0, stream>>>(g) : gather_gemm_kernel<128, 1, 2><<<grid, 256, 0, stream>>>(g);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_conv3d(const sk_conv_src* srcs, int n_src, const void* weight, const float* bias, void* out,
              int B, int ox, int oy, int oz, int cout, int ksize, float* gn_partial,
              void* zeros, void* stream) {
    return conv3d_impl(srcs, n_src, weight, bias, out, B, ox, oy, oz, cout, ksize, gn_partial, zeros, stream, false);
}

int sk_conv3d_box(const sk_conv_src* srcs, int n_src, const void* weight, const float* bias, void* out,
                  int B, int ox, int oy, int oz, int cout, int ksize, float* gn_partial,
                  void* zeros, const int* store_box, void* stream) {
    SK_CHECK_ARG(ksize == 3 || store_box == nullptr, "sk_conv3d_box: a store box needs ksize 3");
    return conv3d_impl(srcs, n_src, weight, bias, out, B, ox, oy, oz, cout, ksize, gn_partial, zeros, stream, false, store_box);
}

int sk_conv3d_box_split(const sk_conv_src* srcs, int n_src, const void* weight, const float* bias, void* out,
                        int B, int ox, int oy, int oz, int cout, int ksize, float* gn_partial,
                        void* zeros, const int* store_box, void* stream) {
    SK_CHECK_ARG(ksize == 3 || store_box == nullptr, "sk_conv3d_box_split: a store box needs ksize 3");
    return conv3d_impl(srcs, n_src, weight, bias, out, B, ox, oy, oz, cout, ksize, gn_partial, zeros, stream, true, store_box);
}

int sk_conv3d_down_act(void* in_raw, const float* affine, const void* weight, const float* bias, void* out, int B,
                       int ox, int oy, int oz, int cin, int cout, float* gn_partial, void* zeros, void* stream) {
    SK_CHECK_ARG(affine, "sk_conv3d_down_act: affine is NULL (use sk_conv3d for an activated input)");
    return launch_down2(in_raw, affine, 1, weight, bias, out, B, ox, oy, oz, cin, cout, gn_partial, zeros, (hipStream_t)stream);
}

// OCP e4m3fn of a float, round to nearest even, saturating at +-448
int64_t sk_conv3d_pack_weight_mix8_host(const float* w, int cout, int cin, void* dst, int* scale_exp) {
    // precision "mix8", 3x3x3, C -> C (C = 32 | 64 | 128): [w_hi as sk_conv3d_pack_weight_host packs it] [fp8 fragments of 2 KiB].
    // b = *scale_exp: the largest power of two with 2^b max|w| <= 240; w_lo = w - fp16(w).
    // C = 32 (conv3_m16_kernel, v_mfma_scale_f32_16x16x128): fragment (tap-row pair rp = 0..4, cout half i, x tap dx): lane l
    // holds, for cout 16 i + (l & 15), the 32 bytes of K block g = l >> 4:  g 0: e4m3(2^(b+11) w_lo) of row 2 rp | g 1:
    // e4m3(2^b w) of row 2 rp | g 2, 3: the same of row 2 rp + 1 (zeros for the tenth row); byte j = input channel j.
    // C = 64 | 128 (conv3_kernel, v_mfma_scale_f32_32x32x64): fragment (((chunk k, row dydz, x tap dx), cout tile nt): lane l
    // holds, for cout 32 nt + (l & 31), K block l >> 5: 0: e4m3(2^(b+11) w_lo), 1: e4m3(2^b w); byte j = input channel 32 k + j.
    if (cout != cin || !(cout == 32 || cout == 64 || cout == 128)) {
        sk::set_error("sk_conv3d_pack_weight_mix8_host: cout = cin = 32 | 64 | 128 (got %d, %d)", cout, cin);
        return SK_ERR_ARG;
    }
    const int64_t f16b = mix8_fp16_bytes(cout, cin);
    const int64_t total = f16b + mix8_fp8_bytes(cout, cin);
    if (!dst) return total;
    const int64_t n = (int64_t)cout * cin * 27;
    std::vector<float> hi(n);
    float wmax = 0.0f;
    for (int64_t i = 0; i < n; ++i) {
        hi[i] = (float)((t16)(w[i]));
        wmax = std::fmax(wmax, std::fabs(w[i]));
    }
    int b = 0;
    while (b < 40 && std::ldexp(wmax, b + 1) <= 240.0f) ++b;
    if (scale_exp) *scale_exp = b;
    const int64_t got = sk_conv3d_pack_weight_host(hi.data(), cout, cin, 3, dst);
    if (got != f16b) return SK_ERR_ARG;
    uint8_t* o = reinterpret_cast<uint8_t*>(dst) + f16b;
    // torch layout (co, ci, dx, dy, dz); row = dy * 3 + dz
    auto W = [&](int co, int ci, int dx, int row) { return (((int64_t)co * cin + ci) * 3 + dx) * 9 + row; };
    auto lo8 = [&](int64_t idx) { return sk::f32_to_e4m3(std::ldexp(w[idx] - hi[idx], b + 11)); };
    auto w8 = [&](int64_t idx) { return sk::f32_to_e4m3(std::ldexp(w[idx], b)); };
    if (cout == 32) {
        for (int rp = 0; rp < 5; ++rp)
            for (int i = 0; i < 2; ++i)
                for (int dx = 0; dx < 3; ++dx) {
                    const int f = (rp * 2 + i) * 3 + dx;
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 32; ++j) {
                            const int g = l >> 4, row = 2 * rp + (g >> 1), co = 16 * i + (l & 15);
                            uint8_t v = 0;
                            if (row < 9) v = (g & 1) ? w8(W(co, j, dx, row)) : lo8(W(co, j, dx, row));
                            o[f * 2048 + l * 32 + j] = v;
                        }
                }
        return total;
    }
    const int NT = cout / 32;
    for (int k = 0; k < cin / 32; ++k)
        for (int row = 0; row < 9; ++row)
            for (int dx = 0; dx < 3; ++dx)
                for (int nt = 0; nt < NT; ++nt) {
                    const int64_t f = ((int64_t)((k * 9 + row) * 3 + dx)) * NT + nt;
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 32; ++j) {
                            const int64_t idx = W(32 * nt + (l & 31), 32 * k + j, dx, row);
                            o[f * 2048 + l * 32 + j] = (l >> 5) ? w8(idx) : lo8(idx);
                        }
                }
    return total;
}

int sk_conv3d_mix8(const sk_conv_src* srcs, int n_src, const void* weight, int weight_scale_exp, const float* bias, void* out,
                   int B, int ox, int oy, int oz, int cout, float* gn_partial, void* zeros, const int* store_box, void* stream) {
    return conv3d_impl(srcs, n_src, weight, bias, out, B, ox, oy, oz, cout, 3, gn_partial, zeros, stream, true, store_box, true,
                       weight_scale_exp);
}

int sk_conv3d_down_act_split(void* in_raw, const float* affine, const void* weight, const float* bias, void* out, int B,
                             int ox, int oy, int oz, int cin, int cout, float* gn_partial, void* zeros, void* stream) {
    SK_CHECK_ARG(affine, "sk_conv3d_down_act_split: affine is NULL (use sk_conv3d_split for an activated input)");
    return launch_down2(in_raw, affine, 1, weight, bias, out, B, ox, oy, oz, cin, cout, gn_partial, zeros, (hipStream_t)stream, true);
}

int sk_conv3d_down_act_mix8(void* in_raw, const float* affine, const void* weight, const float* bias, void* out, int B,
                            int ox, int oy, int oz, int cin, int cout, float* gn_partial, void* zeros, void* stream) {
    SK_CHECK_ARG(affine, "sk_conv3d_down_act_mix8: affine is NULL");
    return launch_down2(in_raw, affine, 2, weight, bias, out, B, ox, oy, oz, cin, cout, gn_partial, zeros, (hipStream_t)stream, true);
}

int sk_conv3d_split(const sk_conv_src* srcs, int n_src, const void* weight, const float* bias, void* out,
                    int B, int ox, int oy, int oz, int cout, int ksize, float* gn_partial,
                    void* zeros, void* stream) {
    return conv3d_impl(srcs, n_src, weight, bias, out, B, ox, oy, oz, cout, ksize, gn_partial, zeros, stream, true);
}

int64_t sk_conv3d_pack_weight_split_host(const float* w, int cout, int cin, int ksize, void* dst) {
    // w = hi + lo with hi = fp16(w), lo = fp16(w - hi): ~22 significant bits.
    // ksize 3: the fragments of an expanded (cout, 3*cin) weight whose 32-channel chunks are [lo_k, hi_k, hi_k] per
    // logical chunk k -- the phase order of the split conv kernel.  ksize 1 / 2: all hi fragments, then all lo fragments.
    if (!(ksize == 1 || ksize == 2 || ksize == 3) || cout % 32 || cin % (ksize == 3 ? 32 : 16)) {
        sk::set_error("sk_conv3d_pack_weight_split_host: unsupported shape cout=%d cin=%d k=%d", cout, cin, ksize);
        return SK_ERR_ARG;
    }
    const int k3 = ksize * ksize * ksize;
    const int64_t n = (int64_t)cout * cin * k3;
    std::vector<float> hi(n), lo(n);
    for (int64_t i = 0; i < n; ++i) {
        const float h = (float)((t16)(w[i]));
        hi[i] = h;
        lo[i] = (float)((t16)(w[i] - h));
    }
    if (ksize != 3) {
        const int64_t half_bytes = sk_conv3d_pack_weight_host(hi.data(), cout, cin, ksize, nullptr);
        if (!dst || half_bytes < 0) return half_bytes < 0 ? half_bytes : 2 * half_bytes;
        sk_conv3d_pack_weight_host(hi.data(), cout, cin, ksize, dst);
        sk_conv3d_pack_weight_host(lo.data(), cout, cin, ksize, (char*)dst + half_bytes);
        return 2 * half_bytes;
    }
    const int cin3 = 3 * cin;
    if (!dst) return sk_conv3d_pack_weight_host(hi.data(), cout, cin3, 3, nullptr);
    std::vector<float> we((size_t)cout * cin3 * 27);
    for (int co = 0; co < cout; ++co)
        for (int k = 0; k < cin / 32; ++k)
            for (int part = 0; part < 3; ++part) {
                const std::vector<float>& src = part == 0 ? lo : hi;
                for (int c = 0; c < 32; ++c)
                    for (int t = 0; t < 27; ++t)
                        we[((size_t)co * cin3 + (3 * k + part) * 32 + c) * 27 + t] = src[((size_t)co * cin + 32 * k + c) * 27 + t];
            }
    return sk_conv3d_pack_weight_host(we.data(), cout, cin3, 3, dst);
}

}  // extern "C"
