// The non-MFMA layers of the U-Net body (oracle/unet_spec.py), all HBM-bound:
//   * stem conv (Cin = 1, 27 taps): normalise the fp16 image exactly as
//     skoots/lib/eval.py:139 does (fp16 sub, fp16 div), conv on the exact-fp32 MFMA;
//   * GroupNorm finalize (deterministic reduction of the conv epilogue partials) and
//     the fused GroupNorm-affine + SiLU pass, in place, 16 B per lane;
//   * heads: 1x1x1 conv to 5 channels + tanh / sigmoid, written in the reference's
//     (B, 5, X, Y, Z) output layout (eval.py:145-147).
#include "common.h"

namespace {

typedef t16 half8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------ stem
// (1) normalise: the B tiles are cut out of the fp16 volume, normalised with the reference's
// fp16 arithmetic (eval.py:139: fp16 sub, fp16 div) and written with an explicit one-voxel
// ZERO frame -- conv zero padding applies to the normalised tile.
// (2)+(3) conv: D[cout][voxel] = W[cout][tap] * patch[tap][voxel] on v_mfma_f32_32x32x16_f16,
// run twice (statistics, then apply): see stem_kernel.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef t16 half4v __attribute__((ext_vector_type(4)));

constexpr int kStemMaxB = 64;   // tiles per launch (the pipeline runs 64: +1 % end to end over 32) (origins travel by value in the kernel arguments)

struct StemArgs {
    const t16* image;   // (X, Y, Z) fp16 volume
    int X, Y, Z;           // volume extents
    int ox[kStemMaxB], oy[kStemMaxB], oz[kStemMaxB];  // tile origins
    int B, Xt, Yt, Zt;     // tile extents
    float mean, stdv;
    const float* weight;   // (27, 32) fp32: [tap = (dx*3+dy)*3+dz][cout]
    const float* bias;     // (32)
    t16* norm;          // workspace (B, Xt+2, Yt+2, Zt+2) fp16
    t16* out;           // (B, Xt, Yt, Zt, 32) fp16 ACTIVATED (apply pass)
    float* partial;        // (B, nblk, 8, 2)     (stats pass)
    const float* affine;   // (B, 2, 32)           (apply pass)
    int nblk;
    int rows;              // y rows per workgroup (stem_rows)
};


__global__ void __launch_bounds__(256) stem_norm_kernel(StemArgs a) {
    // grid (x plane of the framed tile, tile): a block walks one plane with 32-bit index math (one thread per element
    // with three 64-bit divisions each took 50 us per 8 tiles)
    const int b = blockIdx.y, x = blockIdx.x;
    const int py = a.Yt + 2, pz = a.Zt + 2;
    t16* out = a.norm + ((long long)b * (a.Xt + 2) + x) * py * pz;
    const int gx = a.ox[b] + x - 1;
    const bool xin = x >= 1 && x <= a.Xt && gx < a.X;
    const t16* img = a.image + (long long)gx * a.Y * a.Z;
    for (int i = threadIdx.x; i < py * pz; i += 256) {
        const int y = i / pz, z = i - y * pz;
        float v = 0.0f;
        // voxels of a tile that overhang the volume (a tile padded up to a multiple of 4, see unet.py) are zero
        // AFTER normalisation, like the conv frame
        const int gy = a.oy[b] + y - 1, gz = a.oz[b] + z - 1;
        if (xin && y >= 1 && y <= a.Yt && z >= 1 && z <= a.Zt && gy < a.Y && gz < a.Z) {
            const float raw = (float)(img[(long long)gy * a.Z + gz]);
            // eval.py:139  crop.sub(mean).div(std) on an fp16 tensor: each op rounds to fp16
            const float s = (float)((t16)(raw - a.mean));
            v = (float)((t16)(s / a.stdv));
        }
        out[i] = (t16)(v);
    }
}

// One conv pass over the zero-framed normalised tile.  STATS: accumulate the GroupNorm partial
// sums of the fp32 results, store nothing.  !STATS: recompute the same fp32 results, apply the
// GroupNorm affine + SiLU and store the ACTIVATED tensor.  Recomputing is cheaper than a raw
// write + read-modify-write pass: the conv is 4 MFMAs per 32 voxels, the tensor is 64 B/voxel.
// Weights are split w = hi + lo (two fp16 values, 22 significant bits) and the input is exactly
// fp16, so the products are exact and the sums match an fp32 conv to ~1e-7 relative.
// y rows of one x plane per workgroup: up to 150, within 40 KiB of LDS for the three staged planes.  The per-block
// prologue (weight split, tap offsets, staging latency) is amortised over rows*Zt/32 tiles; statistics pass per 8
// tiles of 300x300x20, rocprofv3: 60 rows 189 us, 100: 165, 150: 151, 300: 149 (apply pass 332 / 327 / 324 / 338).
static int stem_rows(int Yt, int Zt) {
    int r = (40 * 1024) / (3 * (Zt + 2) * (int)sizeof(t16)) - 2;
    if (r > 150) r = 150;
#ifdef SK_TUNING
    if (const char* e = getenv("SK_STEM_ROWS")) {
        const int v = atoi(e);
        if (v >= 1 && (size_t)3 * (v + 2) * (Zt + 2) * sizeof(t16) <= 60 * 1024) r = v;
    }
#endif
    if (r > Yt) r = Yt;
    // even split of Yt
    int n = (Yt + r - 1) / r;
    r = (Yt + n - 1) / n;
    return r < 1 ? 1 : r;
}

// MODE 0: statistics only; 1: recompute + affine + SiLU, store the activation; 2 (training, mixed precision):
// statistics AND the raw fp16 result in one pass (the backward needs the raw tensor anyway); 3: as 1, but the
// activation is stored as a split pair [hi (32) | lo (32)] per voxel (value = hi + lo, precision "split"); 4: as 3, but the
// line is [hi fp16 (32) | x8 (32) | lo8 (32)] -- e4m3(16 x) and e4m3(2^15 (x - hi)) -- for a precision "mix8" consumer
template <int MODE>
__global__ void __launch_bounds__(256) stem_kernel(StemArgs a) {
    constexpr bool STATS = MODE == 0 || MODE == 2;
    constexpr int kOutC = (MODE == 3 || MODE == 4) ? 64 : 32;   // halves per output voxel line
    __shared__ float red[4 * 16];
    // per-wave transpose pad (2 KiB): the 32 x 32 (channel, voxel) result tile leaves the MFMA layout as whole 64-byte
    // voxel lines, 16 B per lane -- one store instruction writes 1 KiB contiguous (the tile's 32 voxels are contiguous
    // in the output).  8-byte stores straight from the MFMA layout ran the apply pass at 43 % of the write bandwidth.
    // (MODE 3: 4 KiB per wave -- both halves of the 32 [hi | lo] voxel lines, so that a store instruction writes eight WHOLE
    // 128-byte lines; writing the hi halves and the lo halves in separate instructions left every line half-written between
    // them and ran the split apply pass at 3.7 TB/s against 4.6 for the fp16 one)
    constexpr int kPadW = (MODE == 3 || MODE == 4) ? 4096 : 2048;
    __shared__ __attribute__((aligned(16))) char tpad[MODE == 0 ? 16 : 4 * kPadW];
    extern __shared__ __attribute__((aligned(16))) unsigned int stem_lds[];  // [3][rows+2][Zt+2] halves
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int nyc = (a.Yt + a.rows - 1) / a.rows;
    int blk = blockIdx.x % a.nblk;
    const int b = blockIdx.x / a.nblk;
    const int x = blk / nyc, y0 = (blk % nyc) * a.rows;
    const int rows = min(a.rows, a.Yt - y0);
    const long long nvox = (long long)a.Xt * a.Yt * a.Zt;
    const int py = a.Yt + 2, pz = a.Zt + 2;
    const t16* nb = a.norm + (long long)b * (a.Xt + 2) * py * pz;

    // stage the three padded x planes' rows [y0, y0 + rows + 2): contiguous (rows+2)*pz halves each.  Eight loads per
    // lane are issued before the first LDS store (a plain copy loop chains load -> store, one latency per 256 dwords).
    const int seg_halves = (a.rows + 2) * pz;          // LDS pitch per plane (halves, even)
    const int seg_dw = (rows + 2) * pz / 2;               // dwords to copy (pz is even: Zt % 4 == 0)
    {
        const unsigned int* src[3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
            src[dx] = reinterpret_cast<const unsigned int*>(nb + ((long long)(x + dx) * py + y0) * pz);
        for (int i0 = tid; i0 < seg_dw; i0 += 256 * 4) {
            unsigned int v[3][4];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + 256 * k;
                    v[dx][k] = i < seg_dw ? src[dx][i] : 0u;
                }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + 256 * k;
                    if (i < seg_dw) stem_lds[dx * (seg_halves / 2) + i] = v[dx][k];
                }
        }
    }
    __syncthreads();
    const t16* ls = reinterpret_cast<const t16*>(stem_lds);

    // A operands: lane holds W[cout = l&31][tap = 16m + 8h + j], j = 0..7, m = 0,1 (tap >= 27: 0)
    half8 whi[2], wlo[2];
    int toff[2][8];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int tap = 16 * m + 8 * h + j;
            float wv = tap < 27 ? a.weight[tap * 32 + col] : 0.0f;
            t16 hi = (t16)wv;
            whi[m][j] = hi;
            wlo[m][j] = (t16)(wv - (float)hi);
            int tt = tap < 27 ? tap : 0;  // any valid address: its weight is zero
            toff[m][j] = (tt / 9) * seg_halves + ((tt / 3) % 3) * pz + tt % 3;
        }
    f32x16 binit;
    float ga[16], gb[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4v bv = *reinterpret_cast<const f32x4v*>(a.bias + 8 * q + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            binit[4 * q + j] = bv[j];
            if (!STATS) {
                ga[4 * q + j] = a.affine[(long long)b * 64 + 8 * q + 4 * h + j];
                gb[4 * q + j] = a.affine[(long long)b * 64 + 32 + 8 * q + 4 * h + j];
            }
        }
    }
    float gsum[4] = {0, 0, 0, 0}, gsq[4] = {0, 0, 0, 0};
    const int nloc = rows * a.Zt;                 // voxels of this workgroup
    const int ntile = (nloc + 31) / 32;
    // statistics pass: two tiles per iteration with independent accumulators -- the four MFMAs of a tile form one
    // dependent chain, and a lone chain leaves the matrix pipe waiting on itself
    constexpr int U = MODE == 0 ? 2 : 1;
    for (int t0 = w; t0 < ntile; t0 += 4 * U) {
        half8 b0[U], b1[U];
        bool oku[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = (t0 + 4 * u) * 32 + col;
            oku[u] = i < nloc;
            const int ii = oku[u] ? i : 0;
            const int yl = ii / a.Zt, z = ii - yl * a.Zt;
            const t16* p = ls + yl * pz + z;       // tap (0,0,0) in the staged planes
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                b0[u][j] = p[toff[0][j]];
                b1[u][j] = p[toff[1][j]];
            }
        }
        f32x16 accu[U];
#pragma unroll
        for (int u = 0; u < U; ++u) accu[u] = SK_MFMA_32x32x16_T16(wlo[0], b0[u], binit, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) accu[u] = SK_MFMA_32x32x16_T16(wlo[1], b1[u], accu[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) accu[u] = SK_MFMA_32x32x16_T16(whi[0], b0[u], accu[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) accu[u] = SK_MFMA_32x32x16_T16(whi[1], b1[u], accu[u], 0, 0, 0);
        if (STATS) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (oku[u]) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v0 = accu[u][4 * q], v1 = accu[u][4 * q + 1], v2 = accu[u][4 * q + 2], v3 = accu[u][4 * q + 3];
                        gsum[q] += (v0 + v1) + (v2 + v3);
                        gsq[q] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
                    }
                }
        }
        const int t = t0;              // MODE != 0: one tile per iteration
        const f32x16 acc = accu[0];
        if (MODE != 0) {
            // result values of this lane: channels 8q + 4h + j of voxel `col`; MODE 2 stores them raw
            float r[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (MODE == 2) {
                    r[k] = acc[k];
                } else {
                    const float yv = fmaf(ga[k], acc[k], gb[k]);
                    r[k] = yv * __builtin_amdgcn_rcpf(1.0f + __expf(-yv));
                }
            }
            char* pad = tpad + w * kPadW;
            // the tile's 32 voxels are contiguous in the output: voxel index v0t + c, c = 0..31
            const long long v0t = ((long long)x * a.Yt + y0) * a.Zt + (long long)t * 32;
            t16* ob = a.out + ((long long)b * nvox + v0t) * kOutC;
            if constexpr (MODE == 3 || MODE == 4) {
                // pad: [voxel 32][hi 64 B | lo 64 B], 16-byte chunk c of voxel v at slot c ^ (v & 7); read back as whole lines:
                // lane = (voxel 8 k + (lane >> 3), chunk lane & 7)
#pragma unroll
                for (int part = 0; part < (MODE == 3 ? 2 : 1); ++part)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        half4v hv;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const t16 hi = (t16)r[4 * q + j];
                            hv[j] = part == 0 ? hi : (t16)(r[4 * q + j] - (float)hi);
                        }
                        *reinterpret_cast<half4v*>(pad + col * 128 + (((4 * part + q) ^ (col & 7)) * 16) + 8 * h) = hv;
                    }
                if constexpr (MODE == 4) {   // x8 = e4m3(16 x): chunks 4, 5; lo8 = e4m3(2^15 (x - hi)): chunks 6, 7; byte = channel 8 q + 4 h + j
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float xs[4], ls[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = r[4 * q + j];
                            xs[j] = fminf(fmaxf(v * 16.0f, -448.0f), 448.0f);
                            ls[j] = fminf(fmaxf((v - (float)(t16)v) * 32768.0f, -448.0f), 448.0f);
                        }
                        int px = 0, pl = 0;
                        px = __builtin_amdgcn_cvt_pk_fp8_f32(xs[0], xs[1], px, false);
                        px = __builtin_amdgcn_cvt_pk_fp8_f32(xs[2], xs[3], px, true);
                        pl = __builtin_amdgcn_cvt_pk_fp8_f32(ls[0], ls[1], pl, false);
                        pl = __builtin_amdgcn_cvt_pk_fp8_f32(ls[2], ls[3], pl, true);
                        *reinterpret_cast<int*>(pad + col * 128 + (((4 + (q >> 1)) ^ (col & 7)) * 16) + 8 * (q & 1) + 4 * h) = px;
                        *reinterpret_cast<int*>(pad + col * 128 + (((6 + (q >> 1)) ^ (col & 7)) * 16) + 8 * (q & 1) + 4 * h) = pl;
                    }
                }
                const int lv = lane >> 3, lc = lane & 7;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int vv = 8 * k + lv;
                    const half8 line = *reinterpret_cast<const half8*>(pad + vv * 128 + ((lc ^ (vv & 7)) * 16));
                    if (t * 32 + vv < nloc)
                        __builtin_nontemporal_store(line, reinterpret_cast<half8*>(reinterpret_cast<char*>(ob + (long long)vv * kOutC) + lc * 16));
                }
            } else {
                const int rv = lane >> 2, rc = lane & 3;    // read-back role: voxel (0..15), 16-byte chunk of its line
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    half4v hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) hv[j] = (t16)r[4 * q + j];
                    *reinterpret_cast<half4v*>(pad + col * 64 + ((q ^ ((col >> 1) & 3)) * 16) + 8 * h) = hv;
                }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int vv = rv + 16 * hh;
                    const half8 line = *reinterpret_cast<const half8*>(pad + vv * 64 + ((rc ^ ((vv >> 1) & 3)) * 16));
                    if (t * 32 + vv < nloc)
                        __builtin_nontemporal_store(line, reinterpret_cast<half8*>(reinterpret_cast<char*>(ob + (long long)vv * kOutC) + rc * 16));
                }
            }
        }
    }
    if (STATS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float s = gsum[q], ss = gsq[q];
#pragma unroll
            for (int m = 16; m > 0; m >>= 1) {
                s += __shfl_xor(s, m);
                ss += __shfl_xor(ss, m);
            }
            if (col == 0) {
                int quad = 2 * q + h;
                red[(w * 8 + quad) * 2] = s;
                red[(w * 8 + quad) * 2 + 1] = ss;
            }
        }
        __syncthreads();
        if (tid < 16)
            a.partial[((long long)b * a.nblk + blk) * 16 + tid] =
                red[tid] + red[16 + tid] + red[32 + tid] + red[48 + tid];
    }
}

// ------------------------------------------------------------------------------ GroupNorm
// partial: (B, nblk, C/4, 2) fp32 -> affine (B, 2, C): a = gamma*rstd, b = beta - mean*a
constexpr int kFinThreads = 1024;

// Many rows (a 256^3 training crop leaves 131 072 per sample): one block per sample would read them all.  First
// pass: block g sums rows [g*S, (g+1)*S) in double and stores the result (as float) in ITS OWN first row g*S -- no
// other block reads that row -- then the finalize below walks the rows with stride S.  Fixed order: deterministic.
__global__ void __launch_bounds__(256) gn_prereduce_kernel(float* __restrict__ partial, int nblk, int nv, int S) {
    __shared__ double accv[256];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int r0 = blockIdx.x * S, r1 = min(nblk, r0 + S);
    const int val = tid % nv, sl = tid / nv, nsl = 256 / nv;
    float* base = partial + (long long)b * nblk * nv;
    double s = 0.0;
    for (int k = r0 + sl; k < r1; k += nsl) s += (double)base[(long long)k * nv + val];
    accv[tid] = s;
    __syncthreads();
    if (tid < nv) {
        double a = 0.0;
        for (int k = 0; k < nsl; ++k) a += accv[k * nv + tid];
        base[(long long)r0 * nv + tid] = (float)a;
    }
}

__global__ void __launch_bounds__(kFinThreads) gn_finalize_kernel(const float* __restrict__ partial, int nblk, int stride,
                                                                  int groups, int C, double count,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float eps,
                                                                  float* __restrict__ affine,
                                                                  float* __restrict__ stats) {
    __shared__ double qv[64];            // per channel quad: (sum, sumsq) interleaved, C/4 <= 32
    __shared__ double accv[kFinThreads];
    const int b = blockIdx.x;
    const int nv = C / 2;                // floats per partial row = (C/4) quads x 2
    const int tid = threadIdx.x;
    // thread (value, row-slice): consecutive threads read consecutive floats of a row (coalesced);
    // every value is summed in a fixed order -> deterministic, independent of launch timing
    const int val = tid % nv, sl = tid / nv, nsl = kFinThreads / nv;
    double s = 0.0;
    const float* base = partial + (long long)b * nblk * nv + val;
    const int nrows = (nblk + stride - 1) / stride;
    for (int k = sl; k < nrows; k += nsl) s += (double)base[(long long)k * stride * nv];
    accv[tid] = s;
    __syncthreads();
    if (tid < nv) {
        double a = 0.0;
        for (int k = 0; k < nsl; ++k) a += accv[k * nv + tid];
        qv[tid] = a;
    }
    __syncthreads();
    if (tid < C) {
        const int gs = C / groups;          // channels per group
        const int g = tid / gs;
        const int q0 = g * gs / 4, q1 = (g + 1) * gs / 4;
        double a = 0.0, c = 0.0;
        for (int k = q0; k < q1; ++k) {
            a += qv[2 * k];
            c += qv[2 * k + 1];
        }
        double n = count * gs;
        double mean = a / n;
        double var = c / n - mean * mean;
        if (var < 0.0) var = 0.0;
        float rstd = (float)(1.0 / sqrt(var + (double)eps));
        float ga = gamma[tid] * rstd;
        affine[((long long)b * 2) * C + tid] = ga;
        affine[((long long)b * 2 + 1) * C + tid] = beta[tid] - (float)mean * ga;
        if (stats && tid % gs == 0) {   // (B, groups, 2): mean, rstd -- kept for the backward pass
            stats[((long long)b * groups + g) * 2] = (float)mean;
            stats[((long long)b * groups + g) * 2 + 1] = rstd;
        }
    }
}

// in place: x = silu(a*x + b), 8 channels (16 B) per lane
__global__ void __launch_bounds__(256) gn_silu_kernel(t16* __restrict__ x,
                                                      const float* __restrict__ affine, int C,
                                                      long long nvec_per_batch) {
    const int b = blockIdx.y;
    const int vpc = C / 8;  // 16-byte vectors per voxel
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;  // multiple of vpc (C/8 divides 256)
    const int c0 = (int)(i % vpc) * 8;
    float ga[8], gb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ga[j] = affine[((long long)b * 2) * C + c0 + j];
        gb[j] = affine[((long long)b * 2 + 1) * C + c0 + j];
    }
    half8* p = reinterpret_cast<half8*>(x) + (long long)b * nvec_per_batch;
    for (; i < nvec_per_batch; i += stride) {
        half8 v = __builtin_nontemporal_load(&p[i]);
        half8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float y = fmaf(ga[j], (float)v[j], gb[j]);
            float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-y));
            r[j] = sk::round_t16(y * sg);
        }
        __builtin_nontemporal_store(r, &p[i]);
    }
}

// split tensors: x (B, vox, 2C) = [hi (C) | lo (C)] fp16 per voxel, value = hi + lo; in place, fp32 arithmetic
__global__ void __launch_bounds__(256) gn_silu_split_kernel(t16* __restrict__ x, const float* __restrict__ affine,
                                                            int C, long long nvec_per_batch) {
    const int b = blockIdx.y;
    const int vpc = C / 8;  // 16-byte vectors per voxel and half
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;  // multiple of vpc
    const int c0 = (int)(i % vpc) * 8;
    float ga[8], gb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ga[j] = affine[((long long)b * 2) * C + c0 + j];
        gb[j] = affine[((long long)b * 2 + 1) * C + c0 + j];
    }
    half8* p = reinterpret_cast<half8*>(x) + (long long)b * nvec_per_batch * 2;
    for (; i < nvec_per_batch; i += stride) {
        const long long vox = i / vpc;
        half8* ph = p + vox * (2 * vpc) + (i - vox * vpc);
        half8 vh = ph[0], vl = ph[vpc];
        half8 rh, rl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xv = (float)vh[j] + (float)vl[j];
            const float y = fmaf(ga[j], xv, gb[j]);
            const float sv = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
            rh[j] = sk::round_t16(sv);   // the product rounded to fp32 first: the same bits wherever this activation is fused
            rl[j] = (t16)(sv - (float)rh[j]);
        }
        ph[0] = rh;
        ph[vpc] = rl;
    }
}

// precision "mix8": raw split line [hi (C) | lo16 (C)] in, activated [hi fp16 (C) | per 32-channel chunk: x8 (32 bytes) | lo8 (32 bytes)]
// out, in place (C = 32 | 64 | 128; the line keeps its 4 C bytes).  The four lanes of a 32-channel chunk sit in one wave and
// every lane loads before any lane stores: the x8 / lo8 bytes of one lane overwrite raw lo halves that ANOTHER lane of the same
// chunk has already read.
template <int C>
__global__ void __launch_bounds__(256) gn_silu_mix8_kernel(t16* __restrict__ x, const float* __restrict__ affine, long long nvox_per_batch) {
    constexpr int kOct = C / 8;   // lanes per voxel
    const int b = blockIdx.y;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // (voxel, octet)
    const long long stride = (long long)gridDim.x * 256;           // multiple of kOct
    const int o = (int)(i % kOct);
    float ga[8], gb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ga[j] = affine[((long long)b * 2) * C + 8 * o + j];
        gb[j] = affine[((long long)b * 2 + 1) * C + 8 * o + j];
    }
    char* base = reinterpret_cast<char*>(x) + (long long)b * nvox_per_batch * (4 * C);
    for (; i < nvox_per_batch * kOct; i += stride) {
        char* line = base + (i / kOct) * (4 * C);
        const half8 vh = *reinterpret_cast<const half8*>(line + 16 * o);
        const half8 vl = *reinterpret_cast<const half8*>(line + 2 * C + 16 * o);
        half8 rh;
        float xs[8], ls[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {   // gn_silu_split_kernel's arithmetic up to the split
            const float xv = (float)vh[j] + (float)vl[j];
            const float y = fmaf(ga[j], xv, gb[j]);
            const float sv = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
            rh[j] = sk::round_t16(sv);
            xs[j] = fminf(fmaxf(sv * 16.0f, -448.0f), 448.0f);
            ls[j] = fminf(fmaxf((sv - (float)rh[j]) * 32768.0f, -448.0f), 448.0f);
        }
        int px[2] = {0, 0}, pl[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            px[k] = __builtin_amdgcn_cvt_pk_fp8_f32(xs[4 * k], xs[4 * k + 1], px[k], false);
            px[k] = __builtin_amdgcn_cvt_pk_fp8_f32(xs[4 * k + 2], xs[4 * k + 3], px[k], true);
            pl[k] = __builtin_amdgcn_cvt_pk_fp8_f32(ls[4 * k], ls[4 * k + 1], pl[k], false);
            pl[k] = __builtin_amdgcn_cvt_pk_fp8_f32(ls[4 * k + 2], ls[4 * k + 3], pl[k], true);
        }
        char* part8 = line + 2 * C + 64 * (o >> 2) + 8 * (o & 3);
        *reinterpret_cast<half8*>(line + 16 * o) = rh;
        *reinterpret_cast<int2*>(part8) = make_int2(px[0], px[1]);
        *reinterpret_cast<int2*>(part8 + 32) = make_int2(pl[0], pl[1]);
    }
}

// ------------------------------------------------------------------------------ heads
struct HeadArgs {
    const t16* x;     // (B, n, C) fp16 (activated, or raw when affine != NULL)
    const float* affine; // (B, 2, C) or NULL
    const float* weight; // (5, C)
    const float* bias;   // (5)
    t16* out5;        // (B, 5, n)
    long long n;         // voxels per tile = X*Y*Z
    int C;
    int Y, Z;            // tile extents (X = n / (Y*Z))
    int lx, ly, lz;      // only the box [l, h) of every tile is evaluated (the rest of out5 is not written)
    int bx, by, bz;      // box extents
};

// out5[k][voxel] = act_k( W[k][:] . silu(affine(x[voxel][:])) + bias[k] ) as a 32x32x16 MFMA with
// 5 useful rows: per 32 voxels a wave loads 2 x 16 B per lane, activates 16 values per lane and
// issues 4 MFMAs (weights split hi + lo: exact products, the logits keep fp32 precision).
// SPLIT: x is (B, n, 2C) = [hi | lo] pairs; the (activated) fp32 value is split again into hi + lo MFMA operands
template <int C, bool SPLIT = false>
__global__ void __launch_bounds__(256) heads_kernel(HeadArgs a) {
    static_assert(C == 32, "two K steps of 16");
    constexpr int kLine = C * (SPLIT ? 2 : 1);   // halves per input voxel line
    const int tid = threadIdx.x, lane = tid & 63;
    const int col = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    // A operands: lane holds W[row = col][cin = 16 ks + 8h + j] (rows >= 5 are zero)
    half8 whi[2], wlo[2];
    float ga[2][8], gb[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 16 * ks + 8 * h + j;
            float wv = col < 5 ? a.weight[col * C + c] : 0.0f;
            t16 hi = (t16)wv;
            whi[ks][j] = hi;
            wlo[ks][j] = (t16)(wv - (float)hi);
            ga[ks][j] = a.affine ? a.affine[(long long)b * 2 * C + c] : 1.0f;
            gb[ks][j] = a.affine ? a.affine[(long long)b * 2 * C + C + c] : 0.0f;
        }
    f32x16 binit;
#pragma unroll
    for (int r = 0; r < 16; ++r) binit[r] = 0.0f;
    if (h == 0) {
        binit[0] = a.bias[0];
        binit[1] = a.bias[1];
        binit[2] = a.bias[2];
        binit[3] = a.bias[3];
    } else {
        binit[0] = a.bias[4];  // row 4 = register 0 of the upper half-wave
    }
    const t16* xb = a.x + (long long)b * a.n * kLine;
    t16* ob = a.out5 + (long long)b * 5 * a.n;
    const long long nbox = (long long)a.bx * a.by * a.bz;
    const long long ntiles = (nbox + 31) / 32;
    for (long long t = (long long)blockIdx.x * 4 + (tid >> 6); t < ntiles; t += (long long)gridDim.x * 4) {
        const long long i = t * 32 + col;
        const bool ok = i < nbox;
        const long long ii = ok ? i : 0;
        const int z = (int)(ii % a.bz);
        const long long r2 = ii / a.bz;
        const long long v = ((long long)(a.lx + (int)(r2 / a.by)) * a.Y + (a.ly + (int)(r2 % a.by))) * a.Z + (a.lz + z);
        const half8* p = reinterpret_cast<const half8*>(xb + (ok ? v : 0) * kLine);
        half8 bf[2], bl[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 raw = p[2 * ks + h];
            if constexpr (SPLIT) {
                const half8 rlo = p[C / 8 + 2 * ks + h];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float y = (float)raw[j] + (float)rlo[j];
                    if (a.affine) {
                        y = fmaf(ga[ks][j], y, gb[ks][j]);
                        y = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
                    }
                    raw[j] = (t16)y;
                    bl[ks][j] = (t16)(y - (float)raw[j]);
                }
            } else if (a.affine) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float y = fmaf(ga[ks][j], (float)raw[j], gb[ks][j]);
                    raw[j] = sk::round_t16(y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)));
                }
            }
            bf[ks] = raw;
        }
        f32x16 acc = binit;
        if constexpr (SPLIT) {
            acc = SK_MFMA_32x32x16_T16(whi[0], bl[0], acc, 0, 0, 0);
            acc = SK_MFMA_32x32x16_T16(whi[1], bl[1], acc, 0, 0, 0);
        }
        acc = SK_MFMA_32x32x16_T16(wlo[0], bf[0], acc, 0, 0, 0);
        acc = SK_MFMA_32x32x16_T16(wlo[1], bf[1], acc, 0, 0, 0);
        acc = SK_MFMA_32x32x16_T16(whi[0], bf[0], acc, 0, 0, 0);
        acc = SK_MFMA_32x32x16_T16(whi[1], bf[1], acc, 0, 0, 0);
        if (ok) {
            if (h == 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k)  // tanh(x) = 1 - 2 / (1 + e^{2x})
                    ob[k * a.n + v] = (t16)(1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * acc[k])));
                ob[3 * a.n + v] = (t16)(__builtin_amdgcn_rcpf(1.0f + __expf(-acc[3])));
            } else {
                ob[4 * a.n + v] = (t16)(__builtin_amdgcn_rcpf(1.0f + __expf(-acc[0])));
            }
        }
    }
}

}  // namespace

extern "C" {

int sk_conv3d_stem_num_blocks(int X, int Y, int Z) {
    const int r = stem_rows(Y, Z);
    return X * ((Y + r - 1) / r);
}

static size_t stem_lds_bytes(int Yt, int Zt) { return (size_t)3 * (stem_rows(Yt, Zt) + 2) * (Zt + 2) * sizeof(t16); }

size_t sk_conv3d_stem_workspace_bytes(int B, int Xt, int Yt, int Zt) {
    return (size_t)B * (Xt + 2) * (Yt + 2) * (Zt + 2) * sizeof(t16);
}

static int fill_stem_args(StemArgs& a, const void* image, int X, int Y, int Z, const int32_t* origins_host,
                          int B, int Xt, int Yt, int Zt, float mean, float stdv, const float* weight,
                          const float* bias, int cout, void* workspace, size_t workspace_bytes) {
    SK_CHECK_ARG(image && origins_host && weight && bias && workspace, "sk_conv3d_stem: NULL pointer");
    SK_CHECK_ARG(cout == 32, "sk_conv3d_stem: cout must be 32");
    SK_CHECK_ARG(B >= 1 && B <= kStemMaxB, "sk_conv3d_stem: batch must be in [1,%d]", kStemMaxB);
    SK_CHECK_ARG(stdv != 0.0f, "sk_conv3d_stem: std must be non-zero");
    SK_CHECK_ARG(workspace_bytes >= sk_conv3d_stem_workspace_bytes(B, Xt, Yt, Zt),
                 "sk_conv3d_stem: workspace too small");
    a.image = (const t16*)image;
    a.X = X;
    a.Y = Y;
    a.Z = Z;
    for (int b = 0; b < B; ++b) {
        a.ox[b] = origins_host[3 * b];
        a.oy[b] = origins_host[3 * b + 1];
        a.oz[b] = origins_host[3 * b + 2];
        // a tile may overhang the volume by < 4 voxels per axis (extent padded up to a multiple of 4)
        SK_CHECK_ARG(a.ox[b] >= 0 && a.oy[b] >= 0 && a.oz[b] >= 0 && a.ox[b] + Xt <= X + 3 &&
                         a.oy[b] + Yt <= Y + 3 && a.oz[b] + Zt <= Z + 3 && a.ox[b] < X && a.oy[b] < Y && a.oz[b] < Z,
                     "sk_conv3d_stem: tile %d at (%d,%d,%d)+(%d,%d,%d) outside volume (%d,%d,%d)", b,
                     a.ox[b], a.oy[b], a.oz[b], Xt, Yt, Zt, X, Y, Z);
    }
    a.B = B;
    a.Xt = Xt;
    a.Yt = Yt;
    a.Zt = Zt;
    a.mean = mean;
    a.stdv = stdv;
    a.weight = weight;
    a.bias = bias;
    a.norm = (t16*)workspace;
    a.nblk = sk_conv3d_stem_num_blocks(Xt, Yt, Zt);
    a.rows = stem_rows(Yt, Zt);
    return SK_OK;
}

int sk_conv3d_stem(const void* image, int X, int Y, int Z, const int32_t* origins_host, int B, int Xt,
                   int Yt, int Zt, float mean, float stdv, const float* weight, const float* bias,
                   int cout, float* gn_partial, void* workspace, size_t workspace_bytes, void* stream) {
    SK_CHECK_ARG(gn_partial, "sk_conv3d_stem: gn_partial is NULL");
    StemArgs a{};
    int rc = fill_stem_args(a, image, X, Y, Z, origins_host, B, Xt, Yt, Zt, mean, stdv, weight, bias, cout,
                            workspace, workspace_bytes);
    if (rc) return rc;
    a.partial = gn_partial;
    dim3 g1(Xt + 2, B);
    stem_norm_kernel<<<g1, 256, 0, (hipStream_t)stream>>>(a);
    SK_CHECK_ARG(Zt % 2 == 0 && stem_lds_bytes(Yt, Zt) <= 60 * 1024, "sk_conv3d_stem: tile depth %d unsupported", Zt);
    if (stem_lds_bytes(Yt, Zt) > 40 * 1024)
        SK_CHECK_HIP(hipFuncSetAttribute((const void*)stem_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)stem_lds_bytes(Yt, Zt)));
    stem_kernel<0><<<(unsigned)(a.nblk * B), 256, stem_lds_bytes(Yt, Zt), (hipStream_t)stream>>>(a);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

// One-pass form (round 4, measured against the two-pass form: DESIGN.md section 8): normalise, then the conv ONCE with the
// raw fp16 result stored next to its statistics (stem_kernel<2>, the training path's mode); the consumer (the single-chunk
// 32 -> 32 conv) activates the raw tensor in LDS.
int sk_conv3d_stem_raw(const void* image, int X, int Y, int Z, const int32_t* origins_host, int B, int Xt,
                       int Yt, int Zt, float mean, float stdv, const float* weight, const float* bias,
                       int cout, void* out_raw, float* gn_partial, void* workspace, size_t workspace_bytes, void* stream) {
    SK_CHECK_ARG(gn_partial && out_raw, "sk_conv3d_stem_raw: NULL output");
    StemArgs a{};
    int rc = fill_stem_args(a, image, X, Y, Z, origins_host, B, Xt, Yt, Zt, mean, stdv, weight, bias, cout,
                            workspace, workspace_bytes);
    if (rc) return rc;
    a.partial = gn_partial;
    a.out = (t16*)out_raw;
    dim3 g1(Xt + 2, B);
    stem_norm_kernel<<<g1, 256, 0, (hipStream_t)stream>>>(a);
    SK_CHECK_ARG(Zt % 2 == 0 && stem_lds_bytes(Yt, Zt) <= 60 * 1024, "sk_conv3d_stem_raw: tile depth %d unsupported", Zt);
    if (stem_lds_bytes(Yt, Zt) > 40 * 1024)
        SK_CHECK_HIP(hipFuncSetAttribute((const void*)stem_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)stem_lds_bytes(Yt, Zt)));
    stem_kernel<2><<<(unsigned)(a.nblk * B), 256, stem_lds_bytes(Yt, Zt), (hipStream_t)stream>>>(a);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

static int stem_apply_impl(int B, int Xt, int Yt, int Zt, const float* weight, const float* bias,
                           const float* affine, void* out, int cout, const void* workspace, void* stream, bool split, bool mix8 = false) {
    SK_CHECK_ARG(weight && bias && affine && out && workspace, "sk_conv3d_stem_apply: NULL pointer");
    SK_CHECK_ARG(cout == 32 && B >= 1 && B <= kStemMaxB, "sk_conv3d_stem_apply: bad cout / batch");
    StemArgs a{};
    a.B = B;
    a.Xt = Xt;
    a.Yt = Yt;
    a.Zt = Zt;
    a.weight = weight;
    a.bias = bias;
    a.affine = affine;
    a.norm = (t16*)workspace;
    a.out = (t16*)out;
    a.nblk = sk_conv3d_stem_num_blocks(Xt, Yt, Zt);
    a.rows = stem_rows(Yt, Zt);
    auto kern = mix8 ? stem_kernel<4> : (split ? stem_kernel<3> : stem_kernel<1>);
    if (stem_lds_bytes(Yt, Zt) > 40 * 1024)
        SK_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)stem_lds_bytes(Yt, Zt)));
    kern<<<(unsigned)(a.nblk * B), 256, stem_lds_bytes(Yt, Zt), (hipStream_t)stream>>>(a);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_conv3d_stem_apply(int B, int Xt, int Yt, int Zt, const float* weight, const float* bias,
                         const float* affine, void* out, int cout, const void* workspace, void* stream) {
    return stem_apply_impl(B, Xt, Yt, Zt, weight, bias, affine, out, cout, workspace, stream, false);
}

int sk_conv3d_stem_apply_split(int B, int Xt, int Yt, int Zt, const float* weight, const float* bias,
                               const float* affine, void* out, int cout, const void* workspace, void* stream) {
    return stem_apply_impl(B, Xt, Yt, Zt, weight, bias, affine, out, cout, workspace, stream, true);
}

int sk_conv3d_stem_apply_mix8(int B, int Xt, int Yt, int Zt, const float* weight, const float* bias,
                              const float* affine, void* out, int cout, const void* workspace, void* stream) {
    return stem_apply_impl(B, Xt, Yt, Zt, weight, bias, affine, out, cout, workspace, stream, true, true);
}

// ---- training, mixed precision: the stem as a fast block -------------------------------------------------------
// image (B, X, Y, Z) fp32 -> zero-framed fp16 copy (the MFMA operand; the weights stay exact through the hi + lo split)
static __global__ void __launch_bounds__(256) stem_frame_f32_kernel(const float* __restrict__ image, t16* __restrict__ norm, int B,
                                                            int X, int Y, int Z) {
    const int px = X + 2, py = Y + 2, pz = Z + 2;
    const long long n = (long long)B * px * py * pz;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int z = (int)(i % pz);
        long long t = i / pz;
        const int y = (int)(t % py);
        t /= py;
        const int x = (int)(t % px), b = (int)(t / px);
        float v = 0.0f;
        if (x >= 1 && x <= X && y >= 1 && y <= Y && z >= 1 && z <= Z)
            v = image[(((long long)b * X + (x - 1)) * Y + (y - 1)) * Z + (z - 1)];
        norm[i] = (t16)(v);
    }
}

int sk_train_stem_fwd_f16(const float* image, int B, int X, int Y, int Z, const float* weight_t, const float* bias,
                          void* y16, float* gn_partial, void* workspace, size_t workspace_bytes, void* stream) {
    SK_CHECK_ARG(image && weight_t && bias && y16 && gn_partial && workspace, "sk_train_stem_fwd_f16: NULL pointer");
    SK_CHECK_ARG(B >= 1 && B <= kStemMaxB && X >= 1 && Y >= 1 && Z >= 2 && Z % 2 == 0, "sk_train_stem_fwd_f16: bad extents");
    SK_CHECK_ARG(workspace_bytes >= sk_conv3d_stem_workspace_bytes(B, X, Y, Z), "sk_train_stem_fwd_f16: workspace too small");
    SK_CHECK_ARG(stem_lds_bytes(Y, Z) <= 60 * 1024, "sk_train_stem_fwd_f16: depth %d unsupported", Z);
    StemArgs a{};
    a.B = B;
    a.Xt = X;
    a.Yt = Y;
    a.Zt = Z;
    a.weight = weight_t;
    a.bias = bias;
    a.norm = (t16*)workspace;
    a.out = (t16*)y16;
    a.partial = gn_partial;
    a.nblk = sk_conv3d_stem_num_blocks(X, Y, Z);
    a.rows = stem_rows(Y, Z);
    hipStream_t st = (hipStream_t)stream;
    const long long np = (long long)B * (X + 2) * (Y + 2) * (Z + 2);
    stem_frame_f32_kernel<<<sk::stream_grid(np, 256, 4), 256, 0, st>>>(image, a.norm, B, X, Y, Z);
    SK_CHECK_LAUNCH();
    if (stem_lds_bytes(Y, Z) > 40 * 1024)
        SK_CHECK_HIP(hipFuncSetAttribute((const void*)stem_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)stem_lds_bytes(Y, Z)));
    stem_kernel<2><<<(unsigned)(a.nblk * B), 256, stem_lds_bytes(Y, Z), st>>>(a);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

static int groupnorm_finalize_impl(float* gn_partial, int B, int nblocks, int groups, int C,
                          int64_t voxels, const float* gamma, const float* beta, float eps,
                          float* affine, float* stats, void* stream) {
    SK_CHECK_ARG(gn_partial && gamma && beta && affine, "sk_groupnorm_finalize: NULL pointer");
    SK_CHECK_ARG(C % 4 == 0 && C <= 128 && groups > 0 && C % groups == 0 && (C / groups) % 4 == 0,
                 "sk_groupnorm_finalize: C=%d groups=%d unsupported", C, groups);
    SK_CHECK_ARG(kFinThreads % (C / 2) == 0, "sk_groupnorm_finalize: C/2 must divide %d", kFinThreads);
    int stride = 1;
    if (nblocks > 4096) {  // two passes (see gn_prereduce_kernel); compacts gn_partial in place
        stride = (nblocks + 255) / 256;
        gn_prereduce_kernel<<<dim3((nblocks + stride - 1) / stride, B), 256, 0, (hipStream_t)stream>>>(
            gn_partial, nblocks, C / 2, stride);
    }
    gn_finalize_kernel<<<B, kFinThreads, 0, (hipStream_t)stream>>>(gn_partial, nblocks, stride, groups, C, (double)voxels,
                                                            gamma, beta, eps, affine, stats);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_groupnorm_finalize(float* gn_partial, int B, int nblocks, int groups, int C, int64_t voxels,
                          const float* gamma, const float* beta, float eps, float* affine, void* stream) {
    return groupnorm_finalize_impl(gn_partial, B, nblocks, groups, C, voxels, gamma, beta, eps, affine, nullptr, stream);
}

int sk_groupnorm_finalize_stats(float* gn_partial, int B, int nblocks, int groups, int C, int64_t voxels,
                                const float* gamma, const float* beta, float eps, float* affine, float* stats,
                                void* stream) {
    SK_CHECK_ARG(stats, "sk_groupnorm_finalize_stats: NULL stats");
    return groupnorm_finalize_impl(gn_partial, B, nblocks, groups, C, voxels, gamma, beta, eps, affine, stats, stream);
}

int sk_groupnorm_silu(void* x, const float* affine, int B, int64_t voxels, int C, void* stream) {
    SK_CHECK_ARG(x && affine, "sk_groupnorm_silu: NULL pointer");
    SK_CHECK_ARG(C % 8 == 0 && 256 % (C / 8) == 0, "sk_groupnorm_silu: C=%d unsupported", C);
    long long nvec = voxels * (C / 8);
    unsigned gx = sk::stream_grid(nvec, 256, 4);
    dim3 grid(gx, B);
    gn_silu_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((t16*)x, affine, C, nvec);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_groupnorm_silu_split(void* x, const float* affine, int B, int64_t voxels, int C, void* stream) {
    SK_CHECK_ARG(x && affine, "sk_groupnorm_silu_split: NULL pointer");
    SK_CHECK_ARG(C % 8 == 0 && 256 % (C / 8) == 0, "sk_groupnorm_silu_split: C=%d unsupported", C);
    long long nvec = voxels * (C / 8);
    dim3 grid(sk::stream_grid(nvec, 256, 4), B);
    gn_silu_split_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((t16*)x, affine, C, nvec);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_groupnorm_silu_mix8(void* x, const float* affine, int B, int64_t voxels, int C, void* stream) {
    SK_CHECK_ARG(x && affine, "sk_groupnorm_silu_mix8: NULL pointer");
    SK_CHECK_ARG(C == 32 || C == 64 || C == 128, "sk_groupnorm_silu_mix8: C must be 32, 64 or 128 (got %d)", C);
    dim3 grid(sk::stream_grid(voxels * (C / 8), 256, 4), B);
    auto kern = C == 32 ? gn_silu_mix8_kernel<32> : (C == 64 ? gn_silu_mix8_kernel<64> : gn_silu_mix8_kernel<128>);
    kern<<<grid, 256, 0, (hipStream_t)stream>>>((t16*)x, affine, (long long)voxels);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

static int heads_impl(const void* x, const float* affine, const float* weight, const float* bias, void* out5, int B,
                      int X, int Y, int Z, int C, const int* box_lo_host, const int* box_hi_host, void* stream, bool split) {
    SK_CHECK_ARG(x && weight && bias && out5, "sk_heads: NULL pointer");
    SK_CHECK_ARG(C == 32, "sk_heads: C must be 32");
    SK_CHECK_ARG(X > 0 && Y > 0 && Z > 0, "sk_heads: bad extents");
    HeadArgs a{};
    a.x = (const t16*)x;
    a.affine = affine;
    a.weight = weight;
    a.bias = bias;
    a.out5 = (t16*)out5;
    a.n = (long long)X * Y * Z;
    a.C = C;
    a.Y = Y;
    a.Z = Z;
    a.bx = X;
    a.by = Y;
    a.bz = Z;
    if (box_lo_host && box_hi_host) {
        SK_CHECK_ARG(0 <= box_lo_host[0] && box_lo_host[0] < box_hi_host[0] && box_hi_host[0] <= X &&
                         0 <= box_lo_host[1] && box_lo_host[1] < box_hi_host[1] && box_hi_host[1] <= Y &&
                         0 <= box_lo_host[2] && box_lo_host[2] < box_hi_host[2] && box_hi_host[2] <= Z,
                     "sk_heads: box must be a non-empty box inside the tile");
        a.lx = box_lo_host[0];
        a.ly = box_lo_host[1];
        a.lz = box_lo_host[2];
        a.bx = box_hi_host[0] - a.lx;
        a.by = box_hi_host[1] - a.ly;
        a.bz = box_hi_host[2] - a.lz;
    }
    const long long voxels = (long long)a.bx * a.by * a.bz;
    dim3 grid(sk::stream_grid((voxels + 31) / 32, 4, 4), B);
    if (split)
        heads_kernel<32, true><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    else
        heads_kernel<32><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_heads(const void* x, const float* affine, const float* weight, const float* bias, void* out5, int B,
             int X, int Y, int Z, int C, const int* box_lo_host, const int* box_hi_host, void* stream) {
    return heads_impl(x, affine, weight, bias, out5, B, X, Y, Z, C, box_lo_host, box_hi_host, stream, false);
}

int sk_heads_split(const void* x, const float* affine, const float* weight, const float* bias, void* out5, int B,
                   int X, int Y, int Z, int C, const int* box_lo_host, const int* box_hi_host, void* stream) {
    return heads_impl(x, affine, weight, bias, out5, B, X, Y, Z, C, box_lo_host, box_hi_host, stream, true);
}

}  // extern "C"
