// fp32 precision mode of the U-Net layers: the same graph (oracle/unet_spec.py) with fp32
// activations and fp32 weights on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32
// (bitwise an fmaf chain, 1/16 of the fp16 MFMA rate).
//
// Purpose: parity.  BASELINE.json's north_star asks for embedding / probability tensors within
// 1e-3 of the fp32 reference; fp16 MFMA operands alone move this network's outputs by up to
// ~5e-3 (DESIGN.md "Numerics"), so the fast path can meet that bound only in the RMS sense.
// This mode runs the identical tiling / normalisation / GroupNorm / heads plumbing with fp32
// arithmetic and meets 1e-3 max-abs, which pins every difference of the fast path on operand
// rounding.  It is a gather GEMM (no LDS staging): speed is not its job.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SrcF32 {
    const float* data;  // (B, xs, ys, zs, C) fp32 channels-last, activated
    int C, up;
    int Xs, Ys, Zs;
};

struct ConvF32Args {
    SrcF32 src[2];
    int nsrc;
    const float* w;     // torch layout (cout, cin, k, k, k)
    const float* bias;
    float* out;         // (B, ox, oy, oz, cout)
    float* partial;     // (B, nblk, cout/4, 2) or NULL
    int B, ox, oy, oz, cout, cin, ksize;
    int nblk;
    // data-gradient mode (training): the conv is run with the layer's weight read transposed
    // and tap-flipped in place, rows = original input channels [w_c_lo, w_c_lo + cout)
    int transposed, w_cin_total, w_c_lo, accumulate;
};

// one wave = one 32-voxel column tile x one 32-cout row tile; block = 4 waves = 4 column tiles
__global__ void __launch_bounds__(256) conv_f32_kernel(ConvF32Args a) {
    __shared__ float red[4 * 16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int nct = (a.cout + 31) / 32;
    int blk = blockIdx.x;
    const int ct = blk % nct;       // cout tile
    blk /= nct;
    const int b = blk / a.nblk, vb = blk % a.nblk;
    const long long nvox = (long long)a.ox * a.oy * a.oz;
    const long long v = ((long long)vb * 4 + w) * 32 + col;
    const bool ok = v < nvox;
    const long long vv = ok ? v : 0;
    const int z = (int)(vv % a.oz);
    const long long t = vv / a.oz;
    const int y = (int)(t % a.oy), x = (int)(t / a.oy);
    const int k = a.ksize, stride = (k == 3) ? 1 : k, padw = (k == 3) ? 1 : 0;
    const int co = 32 * ct + col;   // this lane's A row (cout)

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = 32 * ct + 8 * q + 4 * h + j;
            acc[4 * q + j] = (a.bias && c < a.cout) ? a.bias[c] : 0.0f;
        }
    const int k3 = k * k * k;
    int cbase = 0;
    for (int s = 0; s < a.nsrc; ++s) {
        const SrcF32 S = a.src[s];
        const float* sb = S.data + (long long)b * S.Xs * S.Ys * S.Zs * S.C;
        for (int tap = 0; tap < k3; ++tap) {
            const int dx = tap / (k * k), dy = (tap / k) % k, dz = tap % k;
            int xi = x * stride + dx - padw, yi = y * stride + dy - padw, zi = z * stride + dz - padw;
            // bounds are those of the (virtual) full-resolution input
            const int Xf = S.up ? S.Xs * 2 : S.Xs, Yf = S.up ? S.Ys * 2 : S.Ys, Zf = S.up ? S.Zs * 2 : S.Zs;
            const bool inb = xi >= 0 && xi < Xf && yi >= 0 && yi < Yf && zi >= 0 && zi < Zf;
            if (S.up) {
                xi >>= 1;
                yi >>= 1;
                zi >>= 1;
            }
            const float* ip = sb + (((long long)xi * S.Ys + yi) * S.Zs + zi) * S.C;
            for (int m = 0; m < (S.C + 1) / 2; ++m) {
                const int ci = 2 * m + h;  // K index of this lane
                const bool cok = ci < S.C;
                const long long wi = a.transposed
                                         ? ((long long)(cbase + ci) * a.w_cin_total + a.w_c_lo + co) * k3 + (k3 - 1 - tap)
                                         : ((long long)co * a.cin + cbase + ci) * k3 + tap;
                float av = (cok && co < a.cout) ? a.w[wi] : 0.0f;
                float bv = (cok && inb && ok) ? ip[ci] : 0.0f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
        }
        cbase += S.C;
    }
    float gs[4] = {0, 0, 0, 0}, gq[4] = {0, 0, 0, 0};
    if (ok) {
        float* op = a.out + ((long long)b * nvox + v) * a.cout;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int c = 32 * ct + 8 * q + 4 * h + j;
                if (c < a.cout) {
                    float r = acc[4 * q + j];
                    if (a.accumulate) r += op[c];
                    op[c] = r;
                    gs[q] += r;
                    gq[q] += r * r;
                }
            }
    }
    if (a.partial) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float s = gs[q], ss = gq[q];
#pragma unroll
            for (int m = 16; m > 0; m >>= 1) {
                s += __shfl_xor(s, m);
                ss += __shfl_xor(ss, m);
            }
            if (col == 0) {
                red[(w * 8 + 2 * q + h) * 2] = s;
                red[(w * 8 + 2 * q + h) * 2 + 1] = ss;
            }
        }
        __syncthreads();
        if (tid < 16 && 8 * ct * 4 + (tid >> 1) * 4 < a.cout) {
            float tsum = red[tid] + red[16 + tid] + red[32 + tid] + red[48 + tid];
            const int nq = a.cout / 4;
            a.partial[(((long long)b * a.nblk + vb) * nq + 8 * ct) * 2 + tid] = tsum;
        }
    }
}

// Same contraction for sources whose channel counts are multiples of 32 (every layer but the stem and
// the head's data gradient): the weight tile of a (source, 32-channel chunk, dx) stage -- 9 taps x 32 ci
// x 32 cout -- is staged in LDS for the block's 8 column tiles, and a lane reads its activations as
// float4 (K is permuted: lane half h owns channels 8j+4h..8j+4h+3 of the chunk).
constexpr int kWRow = 40;  // LDS row stride (floats): rows 4 apart land on the other 32 banks

__global__ void __launch_bounds__(256) conv_f32_lds_kernel(ConvF32Args a) {
    __shared__ float wl[9 * 32 * kWRow];
    __shared__ float red[8 * 16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int nct = (a.cout + 31) / 32;
    int blk = blockIdx.x;
    const int ct = blk % nct;
    blk /= nct;
    const int nb2 = (a.nblk + 1) / 2;  // blocks of 256 voxels
    const int b = blk / nb2, vb = blk % nb2;
    const long long nvox = (long long)a.ox * a.oy * a.oz;
    const int k = a.ksize, k3 = k * k * k, stride = (k == 3) ? 1 : k, padw = (k == 3) ? 1 : 0;
    const int ntg = (k == 3) ? 3 : 1, tpg = k3 / ntg;  // tap groups x taps per group (9 | 8 | 1)

    int vx[2], vy[2], vz[2];
    bool ok[2];
    long long vidx[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const long long v = ((long long)vb * 8 + 4 * t + w) * 32 + col;
        ok[t] = v < nvox;
        vidx[t] = v;
        const long long vv = ok[t] ? v : 0;
        vz[t] = (int)(vv % a.oz);
        const long long q = vv / a.oz;
        vy[t] = (int)(q % a.oy);
        vx[t] = (int)(q / a.oy);
    }
    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = 32 * ct + 8 * q + 4 * h + j;
            float bv = (a.bias && c < a.cout) ? a.bias[c] : 0.0f;
            acc[0][4 * q + j] = bv;
            acc[1][4 * q + j] = bv;
        }
    int cbase = 0;
    for (int s = 0; s < a.nsrc; ++s) {
        const SrcF32 S = a.src[s];
        const long long svox = (long long)S.Xs * S.Ys * S.Zs;
        const int Xf = S.up ? S.Xs * 2 : S.Xs, Yf = S.up ? S.Ys * 2 : S.Ys, Zf = S.up ? S.Zs * 2 : S.Zs;
        // block-uniform descriptor of the source x-planes this block's 256 consecutive voxels (+ taps) can touch --
        // a window, so that a tensor above 4 GiB (one 512x512x128 tile at 32 channels) stays addressable with 32-bit
        // offsets; masked lanes pass an out-of-range offset and read 0, so no load sits behind a branch
        const long long vfirst = (long long)vb * 256, vlast = min(nvox - 1, vfirst + 255);
        const long long oyz = (long long)a.oy * a.oz;
        int xlo = (int)(vfirst / oyz) * stride - padw, xhi = (int)(vlast / oyz) * stride + (k - 1) - padw;
        xlo = max(xlo, 0);
        xhi = min(xhi, Xf - 1);
        if (S.up) {
            xlo >>= 1;
            xhi >>= 1;
        }
        const long long splane = (long long)S.Ys * S.Zs * S.C;   // floats per source x-plane
        const __amdgpu_buffer_rsrc_t rs = sk::make_rsrc(S.data + (long long)b * svox * S.C + xlo * splane,
                                                        (unsigned)((xhi - xlo + 1) * splane * 4));
        for (int ch = 0; ch < S.C; ch += 32) {
            for (int tg = 0; tg < ntg; ++tg) {
                __syncthreads();
                // stage: wl[tap][ci][co], co contiguous
                for (int e = tid; e < tpg * 32 * 32; e += 256) {
                    const int tl = e % tpg, ci = (e / tpg) % 32, co = e / (tpg * 32);
                    const int tap = tg * tpg + tl;
                    const int cog = 32 * ct + co;
                    float val = 0.0f;
                    if (cog < a.cout) {
                        const long long wi = a.transposed
                                                 ? ((long long)(cbase + ch + ci) * a.w_cin_total + a.w_c_lo + cog) * k3 + (k3 - 1 - tap)
                                                 : ((long long)cog * a.cin + cbase + ch + ci) * k3 + tap;
                        val = a.w[wi];
                    }
                    wl[(tl * 32 + ci) * kWRow + co] = val;
                }
                __syncthreads();
                for (int tl = 0; tl < tpg; ++tl) {
                    const int tap = tg * tpg + tl;
                    const int dx = tap / (k * k), dy = (tap / k) % k, dz = tap % k;
                    const float* wrow = wl + (tl * 32 + 4 * h) * kWRow + col;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        int xi = vx[t] * stride + dx - padw, yi = vy[t] * stride + dy - padw, zi = vz[t] * stride + dz - padw;
                        const bool inb = ok[t] && xi >= 0 && xi < Xf && yi >= 0 && yi < Yf && zi >= 0 && zi < Zf;
                        if (S.up) {
                            xi >>= 1;
                            yi >>= 1;
                            zi >>= 1;
                        }
                        const unsigned off = inb ? ((unsigned)(((xi - xlo) * S.Ys + yi) * S.Zs + zi) * (unsigned)S.C + (unsigned)(ch + 4 * h)) * 4u
                                                 : sk::kOob;
                        f32x4 bq[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) bq[j] = sk::buf_load_f32x4(rs, inb ? off + 32u * j : sk::kOob);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x4 bv = bq[j];
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[(8 * j + u) * kWRow], bv[u], acc[t], 0, 0, 0);
                        }
                    }
                }
            }
        }
        cbase += S.C;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float gs[4] = {0, 0, 0, 0}, gq[4] = {0, 0, 0, 0};
        if (ok[t]) {
            float* op = a.out + ((long long)b * nvox + vidx[t]) * a.cout;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int c = 32 * ct + 8 * q + 4 * h + j;
                    if (c < a.cout) {
                        float r = acc[t][4 * q + j];
                        if (a.accumulate) r += op[c];
                        op[c] = r;
                        gs[q] += r;
                        gq[q] += r * r;
                    }
                }
        }
        if (a.partial) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float s = gs[q], ss = gq[q];
#pragma unroll
                for (int m = 16; m > 0; m >>= 1) {
                    s += __shfl_xor(s, m);
                    ss += __shfl_xor(ss, m);
                }
                if (col == 0) {
                    red[((t * 4 + w) * 8 + 2 * q + h) * 2] = s;
                    red[((t * 4 + w) * 8 + 2 * q + h) * 2 + 1] = ss;
                }
            }
        }
    }
    if (a.partial) {
        __syncthreads();
        // two 128-voxel rows per block, the layout sk_conv3d_f32_num_blocks promises
        if (tid < 32) {
            const int t = tid >> 4, e = tid & 15;
            const int row = 2 * vb + t;
            if (row < a.nblk && 8 * ct * 4 + (e >> 1) * 4 < a.cout) {
                const float* r0 = red + t * 64;
                float tsum = r0[e] + r0[16 + e] + r0[32 + e] + r0[48 + e];
                const int nq = a.cout / 4;
                a.partial[(((long long)b * a.nblk + row) * nq + 8 * ct) * 2 + e] = tsum;
            }
        }
    }
}

// Data gradient of a k=2, stride-2 conv: dX[2c + p][ci] = sum_co W[co][ci][p] * dY[c][co].  The
// weight tap depends on the parity p of the fine voxel, so a wave takes 32 fine voxels of ONE parity
// class (blockIdx.y): column = coarse voxel, K = the layer's output channels.
struct ConvT2Args {
    const float* dy;   // (B, cx, cy, cz, K)
    const float* w;    // (K, cin, 2, 2, 2)
    float* dx;         // (B, 2cx, 2cy, 2cz, cin)
    int B, cx, cy, cz, K, cin, nblk, accumulate;
};

__global__ void __launch_bounds__(256) conv_t2_f32_kernel(ConvT2Args a) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int nct = (a.cin + 31) / 32;
    int blk = blockIdx.x;
    const int ct = blk % nct;
    blk /= nct;
    const int b = blk / a.nblk, vb = blk % a.nblk;
    const int p = blockIdx.y, px = p >> 2, py = (p >> 1) & 1, pz = p & 1;
    const long long nvox = (long long)a.cx * a.cy * a.cz;
    const long long v = ((long long)vb * 4 + w) * 32 + col;
    const bool ok = v < nvox;
    const long long vv = ok ? v : 0;
    const int z = (int)(vv % a.cz);
    const long long t = vv / a.cz;
    const int y = (int)(t % a.cy), x = (int)(t / a.cy);
    const int co = 32 * ct + col;  // A row: the layer's INPUT channel
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const float* ip = a.dy + ((long long)b * nvox + vv) * a.K;
    for (int m = 0; m < (a.K + 1) / 2; ++m) {
        const int k = 2 * m + h;
        const bool kok = k < a.K;
        float av = (kok && co < a.cin) ? a.w[((long long)k * a.cin + co) * 8 + p] : 0.0f;
        float bv = (kok && ok) ? ip[k] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    if (ok) {
        const long long fi = (((long long)b * 2 * a.cx + 2 * x + px) * 2 * a.cy + 2 * y + py) * 2 * a.cz + 2 * z + pz;
        float* op = a.dx + fi * a.cin;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int c = 32 * ct + 8 * q + 4 * h + j;
                if (c < a.cin) {
                    float r = acc[4 * q + j];
                    if (a.accumulate) r += op[c];
                    op[c] = r;
                }
            }
    }
}

__global__ void __launch_bounds__(256) gn_silu_f32_kernel(float* __restrict__ x, const float* __restrict__ affine,
                                                          int C, long long n_per_batch) {
    const int b = blockIdx.y;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    float* p = x + (long long)b * n_per_batch;
    for (; i < n_per_batch; i += stride) {
        const int c = (int)(i % C);
        float y = fmaf(affine[(long long)b * 2 * C + c], p[i], affine[(long long)b * 2 * C + C + c]);
        p[i] = y / (1.0f + expf(-y));
    }
}

// First conv of the network in fp32 (Cin = 1, 27 taps, Cout = 32; the training forward and the fp32 parity mode).  The
// generic kernel above spends one scalar load pair and one half-used MFMA per tap (2.7 ms at 256^3).  Here a block
// stages the three x planes' rows [y0 - 1, y0 + rows] in LDS (zero frame included), a wave tile = 32 voxels x 32 couts
// takes 14 v_mfma_f32_32x32x2f32 (two taps per instruction: lane half h holds tap 2m + h), the B values are ds_read_b32
// at per-lane precomputed tap offsets, the weights sit in 14 registers.  rows*Z is a multiple of 128 and the plane
// starts are 128-aligned, so a block covers whole rows of the (nvox / 128) GroupNorm partial table: it writes its sums
// into the first and zeros into the others.
struct StemF32Args {
    const float* x;      // (B, X, Y, Z)
    const float* w;      // (32, 1, 3, 3, 3)
    const float* bias;   // (32)
    float* out;          // (B, X, Y, Z, 32) raw
    float* partial;      // (B, nblk, 8, 2) or NULL
    int B, X, Y, Z, rows, nblk;
};

__global__ void __launch_bounds__(256) stem_f32_kernel(StemF32Args a) {
    __shared__ float red[4 * 16];
    extern __shared__ __attribute__((aligned(16))) float stem32_lds[];  // [3][rows + 2][Z + 2]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int nyc = (a.Y + a.rows - 1) / a.rows;
    int blk = blockIdx.x;
    const int yc = blk % nyc;
    blk /= nyc;
    const int x = blk % a.X, b = blk / a.X;
    const int y0 = yc * a.rows, rows = min(a.rows, a.Y - y0);
    const int pz = a.Z + 2, seg = (a.rows + 2) * pz;
    const float* xb = a.x + (long long)b * a.X * a.Y * a.Z;
    // stage: plane dx, row r (y = y0 - 1 + r), column c (z = c - 1); zero outside the volume
    for (int i = tid; i < 3 * (rows + 2) * pz; i += 256) {
        const int dx = i / ((rows + 2) * pz), rem = i - dx * (rows + 2) * pz;
        const int r = rem / pz, c = rem - r * pz;
        const int xi = x + dx - 1, yi = y0 - 1 + r, zi = c - 1;
        float v = 0.0f;
        if (xi >= 0 && xi < a.X && yi >= 0 && yi < a.Y && zi >= 0 && zi < a.Z)
            v = xb[((long long)xi * a.Y + yi) * a.Z + zi];
        stem32_lds[dx * seg + r * pz + c] = v;
    }
    float wreg[14];
    int toff[14];
#pragma unroll
    for (int m = 0; m < 14; ++m) {
        const int tap = 2 * m + h;
        wreg[m] = tap < 27 ? a.w[col * 27 + tap] : 0.0f;   // A row = cout = col
        const int tt = tap < 27 ? tap : 0;
        toff[m] = (tt / 9) * seg + ((tt / 3) % 3) * pz + tt % 3;
    }
    f32x16 binit;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) binit[4 * q + j] = a.bias ? a.bias[8 * q + 4 * h + j] : 0.0f;
    __syncthreads();
    float gs[4] = {0, 0, 0, 0}, gq[4] = {0, 0, 0, 0};
    const int nloc = rows * a.Z, ntile = (nloc + 31) / 32;
    const long long v0 = ((long long)x * a.Y + y0) * a.Z;       // first voxel of the block (in-sample index)
    float* ob = a.out + ((long long)b * a.X * a.Y * a.Z + v0) * 32;
    for (int t = w; t < ntile; t += 4) {
        const int i = t * 32 + col;
        const bool ok = i < nloc;
        const int ii = ok ? i : 0;
        const int yl = ii / a.Z, z = ii - yl * a.Z;
        const float* p = stem32_lds + yl * pz + z;
        f32x16 acc = binit;
#pragma unroll
        for (int m = 0; m < 14; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[m], p[toff[m]], acc, 0, 0, 0);
        if (ok) {
            float* op = ob + (long long)i * 32 + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 r = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                *reinterpret_cast<f32x4*>(op + 8 * q) = r;
                gs[q] += (r[0] + r[1]) + (r[2] + r[3]);
                gq[q] += (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
            }
        }
    }
    if (a.partial) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float s = gs[q], ss = gq[q];
#pragma unroll
            for (int m = 16; m > 0; m >>= 1) {
                s += __shfl_xor(s, m);
                ss += __shfl_xor(ss, m);
            }
            if (col == 0) {
                red[(w * 8 + 2 * q + h) * 2] = s;
                red[(w * 8 + 2 * q + h) * 2 + 1] = ss;
            }
        }
        __syncthreads();
        // partial rows [v0 / 128, (v0 + nloc) / 128) belong to this block
        const long long r0 = v0 / 128;
        const int nrow = nloc / 128;
        float* pr = a.partial + ((long long)b * a.nblk + r0) * 16;
        if (tid < 16) pr[tid] = red[tid] + red[16 + tid] + red[32 + tid] + red[48 + tid];
        for (int i = 16 + tid; i < nrow * 16; i += 256) pr[i] = 0.0f;
    }
}

// Data gradient of a pointwise conv with FEW output channels (the heads: K = 5 logits -> 32 features): HBM-bound
// (20 B in, 128 B out per voxel), so no matrix instruction -- a lane owns 4 consecutive input channels of a voxel
// (K x 4 weights in registers), 16-byte loads of dy are shared by the voxel's lanes, one 16-byte store per lane.
template <int K>
__global__ void __launch_bounds__(256) pointwise_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              float* __restrict__ dx, long long nvox, int cin_total,
                                                              int c_lo, int c_n, int accumulate) {
    const int nq = c_n / 4;                       // lanes per voxel
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int q = (int)(i % nq);                  // fixed per lane: 256 and the grid stride are multiples of nq
    float wk[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) wk[k][j] = w[(long long)k * cin_total + c_lo + 4 * q + j];  // (K, cin, 1, 1, 1)
    for (; i < nvox * nq; i += stride) {
        const long long v = i / nq;
        float g[K];
#pragma unroll
        for (int k = 0; k < K; ++k) g[k] = dy[v * K + k];
        float4 r = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < K; ++k) {
            r.x = fmaf(g[k], wk[k][0], r.x);
            r.y = fmaf(g[k], wk[k][1], r.y);
            r.z = fmaf(g[k], wk[k][2], r.z);
            r.w = fmaf(g[k], wk[k][3], r.w);
        }
        float4* o = reinterpret_cast<float4*>(dx + v * c_n + 4 * q);
        if (accumulate) {
            const float4 p = *o;
            r = {p.x + r.x, p.y + r.y, p.z + r.z, p.w + r.w};
        }
        *o = r;
    }
}

}  // namespace

// every source a multiple of 32 channels -> LDS-staged kernel, else the plain gather kernel
// rows per block of stem_f32_kernel, or 0 when the shape does not fit its partial-row scheme
static int stem_f32_rows(int Y, int Z) {
    if (((long long)Y * Z) % 128) return 0;
    int gran = 1;
    while ((gran * Z) % 128) gran *= 2;                       // rows * Z must be a multiple of 128
    int r = (40 * 1024) / (3 * (Z + 2) * 4) - 2;              // three staged planes within 40 KiB
    r = r / gran * gran;
    if (r > 64) r = 64 / gran * gran;
    if (r < gran || Y % gran) return 0;
    return r;
}

static int launch_conv_f32(const ConvF32Args& a, hipStream_t st) {
    if (a.nsrc == 1 && a.src[0].C == 1 && a.ksize == 3 && a.cout == 32 && !a.transposed && !a.accumulate && !a.src[0].up) {
        const int rows = stem_f32_rows(a.oy, a.oz);
        if (rows > 0) {
            StemF32Args s{};
            s.x = a.src[0].data;
            s.w = a.w;
            s.bias = a.bias;
            s.out = a.out;
            s.partial = a.partial;
            s.B = a.B;
            s.X = a.ox;
            s.Y = a.oy;
            s.Z = a.oz;
            s.rows = rows;
            s.nblk = a.nblk;
            const size_t lds = (size_t)3 * (rows + 2) * (a.oz + 2) * sizeof(float);
            const unsigned grid = (unsigned)(a.B * a.ox * ((a.oy + rows - 1) / rows));
            stem_f32_kernel<<<grid, 256, lds, st>>>(s);
            return SK_OK;
        }
    }
    bool lds = true;
    for (int i = 0; i < a.nsrc; ++i) lds = lds && (a.src[i].C % 32 == 0);
    for (int i = 0; i < a.nsrc && lds; ++i)
        // the kernel addresses a window of x-planes per block with 32-bit offsets: 256 voxels + the taps
        SK_CHECK_ARG((long long)a.src[i].Ys * a.src[i].Zs * a.src[i].C * 4 * (256 / ((long long)a.oy * a.oz) + 6) < (1LL << 32),
                     "fp32 conv: source %d: x-planes too large for 32-bit window offsets", i);
    const int nct = (a.cout + 31) / 32;
    if (lds) {
        unsigned grid = (unsigned)(((a.nblk + 1) / 2) * a.B * nct);
        conv_f32_lds_kernel<<<grid, 256, 0, st>>>(a);
    } else {
        unsigned grid = (unsigned)(a.nblk * a.B * nct);
        conv_f32_kernel<<<grid, 256, 0, st>>>(a);
    }
    return SK_OK;
}

extern "C" {

int sk_conv3d_f32_num_blocks(int ox, int oy, int oz) {
    return (int)(((long long)ox * oy * oz + 127) / 128);
}

int sk_conv3d_f32(const sk_conv_src* srcs, int n_src, const float* weight, const float* bias, float* out,
                  int B, int ox, int oy, int oz, int cout, int ksize, float* gn_partial, void* stream) {
    SK_CHECK_ARG(srcs && weight && bias && out, "sk_conv3d_f32: NULL pointer");
    SK_CHECK_ARG(n_src == 1 || n_src == 2, "sk_conv3d_f32: n_src must be 1 or 2");
    SK_CHECK_ARG(ksize == 1 || ksize == 2 || ksize == 3, "sk_conv3d_f32: ksize must be 1, 2 or 3");
    SK_CHECK_ARG(cout >= 1 && (gn_partial == nullptr || cout % 32 == 0),
                 "sk_conv3d_f32: GroupNorm partials need cout %% 32 == 0");
    ConvF32Args a{};
    a.nsrc = n_src;
    for (int i = 0; i < n_src; ++i) {
        SK_CHECK_ARG(srcs[i].data && srcs[i].c > 0 && srcs[i].affine == nullptr, "sk_conv3d_f32: bad source %d", i);
        int up = srcs[i].upsample ? 1 : 0;
        SK_CHECK_ARG(!up || (ksize == 3 && ox % 2 == 0 && oy % 2 == 0 && oz % 2 == 0),
                     "sk_conv3d_f32: upsampled source needs ksize 3 and even output extents");
        a.src[i].data = (const float*)srcs[i].data;
        a.src[i].C = srcs[i].c;
        a.src[i].up = up;
        int s = (ksize == 3) ? 1 : ksize;
        a.src[i].Xs = up ? ox / 2 : ox * s;
        a.src[i].Ys = up ? oy / 2 : oy * s;
        a.src[i].Zs = up ? oz / 2 : oz * s;
        a.cin += srcs[i].c;
    }
    a.w = weight;
    a.bias = bias;
    a.out = out;
    a.partial = gn_partial;
    a.B = B;
    a.ox = ox;
    a.oy = oy;
    a.oz = oz;
    a.cout = cout;
    a.ksize = ksize;
    a.nblk = sk_conv3d_f32_num_blocks(ox, oy, oz);
    if (int rc = launch_conv_f32(a, (hipStream_t)stream)) return rc;
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_train_conv_dgrad(const float* dy, const float* weight, float* dx, int B, int ox, int oy, int oz,
                        int cout, int cin_total, int cin_lo, int cin_n, int ksize, int accumulate, void* stream) {
    SK_CHECK_ARG(dy && weight && dx, "sk_train_conv_dgrad: NULL pointer");
    SK_CHECK_ARG(ksize == 1 || ksize == 2 || ksize == 3, "sk_train_conv_dgrad: ksize must be 1, 2 or 3");
    SK_CHECK_ARG(cout >= 1 && cin_lo >= 0 && cin_n >= 1 && cin_lo + cin_n <= cin_total,
                 "sk_train_conv_dgrad: bad channel range");
    SK_CHECK_ARG(B >= 1 && ox >= 1 && oy >= 1 && oz >= 1, "sk_train_conv_dgrad: bad extents");
    if (ksize == 2) {
        SK_CHECK_ARG(cin_lo == 0 && cin_n == cin_total, "sk_train_conv_dgrad: ksize 2 takes the whole input");
        ConvT2Args a{};
        a.dy = dy;
        a.w = weight;
        a.dx = dx;
        a.B = B;
        a.cx = ox;
        a.cy = oy;
        a.cz = oz;
        a.K = cout;
        a.cin = cin_total;
        a.nblk = sk_conv3d_f32_num_blocks(ox, oy, oz);
        a.accumulate = accumulate ? 1 : 0;
        dim3 grid((unsigned)(a.nblk * B * ((cin_total + 31) / 32)), 8);
        conv_t2_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
        SK_CHECK_LAUNCH();
        return SK_OK;
    }
    if (ksize == 1 && cout == 5 && cin_n % 4 == 0 && 256 % (cin_n / 4) == 0 && ((uintptr_t)dx % 16 == 0)) {
        const long long nvox = (long long)B * ox * oy * oz;
        const unsigned grid = sk::stream_grid(nvox * (cin_n / 4), 256, 4);
        pointwise_dgrad_kernel<5><<<grid, 256, 0, (hipStream_t)stream>>>(dy, weight, dx, nvox, cin_total, cin_lo, cin_n,
                                                                        accumulate ? 1 : 0);
        SK_CHECK_LAUNCH();
        return SK_OK;
    }
    ConvF32Args a{};
    a.nsrc = 1;
    a.src[0].data = dy;
    a.src[0].C = cout;
    a.src[0].up = 0;
    a.src[0].Xs = ox;
    a.src[0].Ys = oy;
    a.src[0].Zs = oz;
    a.cin = cout;
    a.w = weight;
    a.bias = nullptr;
    a.out = dx;
    a.partial = nullptr;
    a.B = B;
    a.ox = ox;
    a.oy = oy;
    a.oz = oz;
    a.cout = cin_n;
    a.ksize = ksize;
    a.nblk = sk_conv3d_f32_num_blocks(ox, oy, oz);
    a.transposed = 1;
    a.w_cin_total = cin_total;
    a.w_c_lo = cin_lo;
    a.accumulate = accumulate ? 1 : 0;
    if (int rc = launch_conv_f32(a, (hipStream_t)stream)) return rc;
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_groupnorm_silu_f32(float* x, const float* affine, int B, int64_t voxels, int C, void* stream) {
    SK_CHECK_ARG(x && affine && C > 0, "sk_groupnorm_silu_f32: bad arguments");
    long long n = voxels * C;
    dim3 grid(sk::stream_grid(n, 256, 4), B);
    gn_silu_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, affine, C, n);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

}  // extern "C"
