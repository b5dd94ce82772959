// Config-5 training step (BASELINE.json configs[4]; reference skoots/train/engine.py:456-499):
// forward of the U-Net -> three Tversky losses (train/loss.py:157-209) on the probability map, the
// skeleton map and the baked-skeleton embedding probability (lib/embedding_to_prob.py:5-51 over
// lib/vector_to_embedding.py:79-105) -> backward -> AdamW.
//
// Everything here is fp32 on channels-last (B, voxels, C) tensors.  The forward reuses the fp32
// layer kernels (conv3d_f32.hip); this file holds what only training needs:
//   * out-of-place GroupNorm-affine + SiLU (the raw conv output is kept for the backward pass),
//   * the GroupNorm + SiLU backward (one reduction pass, one apply pass),
//   * the fused loss: one reduction pass over the logits for all three Tversky terms, one pass
//     that writes d(loss)/d(logits),
//   * the weight gradient of a conv as a voxel-reduction GEMM on v_mfma_f32_32x32x2_f32,
//   * 2x2x2 sum pooling (backward of the nearest upsampling), AdamW.
// All reductions are two-stage (per-block partials, then a fixed-order sum in double), so a step
// is deterministic.
#include "common.h"

#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------
// GroupNorm affine + SiLU, forward, out of place
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gn_silu_fwd_kernel(const float* __restrict__ y, const float* __restrict__ affine,
                                                          float* __restrict__ z, int C, long long n_per_batch) {
    const int b = blockIdx.y;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    const float* p = y + (long long)b * n_per_batch;
    float* q = z + (long long)b * n_per_batch;
    for (; i < n_per_batch; i += stride) {
        const int c = (int)(i % C);
        float u = fmaf(affine[(long long)b * 2 * C + c], p[i], affine[(long long)b * 2 * C + C + c]);
        q[i] = u / (1.0f + expf(-u));
    }
}

// ------------------------------------------------------------------------------------------
// GroupNorm + SiLU backward.  z = silu(u), u = gamma*xh + beta, xh = (y - mean_g) * rstd_g.
//   du = dz * silu'(u);  S1[b,c] = sum_v du,  S2[b,c] = sum_v du*xh
//   dgamma = sum_b S2, dbeta = sum_b S1
//   dy = rstd * (gamma*du - m1 - xh*m2),  m1 = sum_{c in g} gamma*S1 / n,  m2 = sum_{c in g} gamma*S2 / n
// ------------------------------------------------------------------------------------------
typedef t16 half8_t __attribute__((ext_vector_type(8)));
// voxels per reduction block: about 65536 / C blocks per sample (2048 for 32 channels; the one-block finalize reads
// blocks x 4C floats, so wide layers get fewer), but never fewer than 512 voxels a block -- a quarter-resolution layer
// (262 k voxels, 128 channels) still gets 512 blocks (with a fixed 8192 voxels it got 32)
__host__ __device__ inline int gnb_vox(long long voxels, int C) {
    const long long target = 65536 / (C < 32 ? 32 : C);
    long long v = (voxels + target - 1) / target;
    v = (v + 511) / 512 * 512;
    return (int)(v < 512 ? 512 : v);
}
__host__ __device__ inline int gnb_blocks(long long voxels, int C) {
    const int vpb = gnb_vox(voxels, C);
    return (int)((voxels + vpb - 1) / vpb);
}

// mixed-precision kernels: hardware exp2 / rcp (relative error ~1e-6, far below the fp16 operands they feed); the
// fp32 parity kernels keep expf and the IEEE division
__device__ inline float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ inline float silu_grad_fast(float u) {
    const float s = sigmoid_fast(u);
    return s * (1.0f + u * (1.0f - s));
}

__device__ inline float silu_grad(float u) {
    float s = sigmoid_(u);
    return s * (1.0f + u * (1.0f - s));
}

__global__ void __launch_bounds__(256) gn_bwd_reduce_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                            const float* __restrict__ affine,
                                                            const float* __restrict__ stats, int C, int groups,
                                                            long long voxels, int nblk, float* __restrict__ partial) {
    __shared__ float red[256 * 2];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int c = tid % C, row = tid / C, rows = 256 / C;
    const int g = c / (C / groups);
    const float a = affine[(long long)b * 2 * C + c], bb = affine[(long long)b * 2 * C + C + c];
    const float mean = stats[((long long)b * groups + g) * 2], rstd = stats[((long long)b * groups + g) * 2 + 1];
    const int vpb = gnb_vox(voxels, C);
    const long long v0 = (long long)blockIdx.x * vpb;
    long long v1 = v0 + vpb;
    if (v1 > voxels) v1 = voxels;
    float s1 = 0.0f, s2 = 0.0f;
    for (long long v = v0 + row; v < v1; v += rows) {
        const long long i = ((long long)b * voxels + v) * C + c;
        const float yv = y[i];
        const float du = dz[i] * silu_grad(fmaf(a, yv, bb));
        s1 += du;
        s2 += du * ((yv - mean) * rstd);
    }
    red[tid * 2] = s1;
    red[tid * 2 + 1] = s2;
    __syncthreads();
    if (tid < C) {
        for (int r = 1; r < rows; ++r) {
            s1 += red[(r * C + tid) * 2];
            s2 += red[(r * C + tid) * 2 + 1];
        }
        float* o = partial + (((long long)b * nblk + blockIdx.x) * C + tid) * 2;
        o[0] = s1;
        o[1] = s2;
    }
}

// one block; coef (B, C, 3) = rstd*gamma, rstd*m1, rstd*m2; dgamma/dbeta (C) overwritten
__global__ void __launch_bounds__(1024) gn_bwd_finalize_kernel(const float* __restrict__ partial, int B, int nblk,
                                                               int C, int groups, double voxels,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ stats,
                                                               float* __restrict__ coef, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
    __shared__ double accv[1024];
    __shared__ double sums[256];  // (C, 2)
    __shared__ double gm[2 * 16];
    const int tid = threadIdx.x;
    const int nv = 2 * C;  // floats per partial row
    const int val = tid % nv, sl = tid / nv, nsl = 1024 / nv;
    double dg = 0.0, db = 0.0;
    for (int b = 0; b < B; ++b) {
        double s = 0.0;
        const float* base = partial + (long long)b * nblk * nv + val;
        if (sl < nsl)
            for (int k = sl; k < nblk; k += nsl) s += (double)base[(long long)k * nv];
        accv[tid] = s;
        __syncthreads();
        if (tid < nv) {
            double t = 0.0;
            for (int k = 0; k < nsl; ++k) t += accv[k * nv + tid];
            sums[tid] = t;
        }
        __syncthreads();
        const int gs = C / groups;
        if (tid < groups) {
            double m1 = 0.0, m2 = 0.0;
            for (int c = tid * gs; c < (tid + 1) * gs; ++c) {
                m1 += (double)gamma[c] * sums[2 * c];
                m2 += (double)gamma[c] * sums[2 * c + 1];
            }
            const double n = voxels * gs;
            gm[2 * tid] = m1 / n;
            gm[2 * tid + 1] = m2 / n;
        }
        __syncthreads();
        if (tid < C) {
            const int g = tid / gs;
            const float rstd = stats[((long long)b * groups + g) * 2 + 1];
            float* o = coef + ((long long)b * C + tid) * 3;
            o[0] = rstd * gamma[tid];
            o[1] = (float)((double)rstd * gm[2 * g]);
            o[2] = (float)((double)rstd * gm[2 * g + 1]);
            db += sums[2 * tid];
            dg += sums[2 * tid + 1];
        }
        __syncthreads();
    }
    if (tid < C) {
        dgamma[tid] = (float)dg;
        dbeta[tid] = (float)db;
    }
}

__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                           const float* __restrict__ affine,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ coef, float* __restrict__ dy,
                                                           int C, int groups, long long n_per_batch) {
    const int b = blockIdx.y;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    const long long off = (long long)b * n_per_batch;
    for (; i < n_per_batch; i += stride) {
        const int c = (int)(i % C);
        const int g = c / (C / groups);
        const float a = affine[(long long)b * 2 * C + c], bb = affine[(long long)b * 2 * C + C + c];
        const float mean = stats[((long long)b * groups + g) * 2], rstd = stats[((long long)b * groups + g) * 2 + 1];
        const float* k = coef + ((long long)b * C + c) * 3;
        const float yv = y[off + i];
        const float du = dz[off + i] * silu_grad(fmaf(a, yv, bb));
        dy[off + i] = k[0] * du - k[1] - ((yv - mean) * rstd) * k[2];
    }
}

// ------------------------------------------------------------------------------------------
// Fused loss.  logits (B, n, 5): [0:3] -> tanh = vectors, [3] -> sigmoid = skeleton, [4] -> sigmoid =
// probability (engine.py:461-463).  E = index + v*scale (vector_to_embedding.py:104-105, N = 1),
// pe = exp(sum_k (E_k - S_k)^2 / (-2 (sigma_k + 1e-16)^2))  (embedding_to_prob.py:38-49).
// Per sample and per loss: TP = sum p*g, FPs = sum p*(1-g), FNs = sum (1-p)*g  (loss.py:189-193).
// ------------------------------------------------------------------------------------------
constexpr int kLossSums = 11;  // 3 losses x (TP, FPs, FNs) + foreground counts of the two targets

struct LossArgs {
    const float* logits;  // (B, n, 5)
    const float* masks;   // (B, n)   instance ids (> 0 = foreground)
    const float* skel;    // (B, n)   skeleton mask (> 0)
    const float* baked;   // (B, 3, n)
    long long n;
    int Y, Z;
    float scale[3];
    float inv_var[3];     // 1 / (-2 (sigma+1e-16)^2)
};

__device__ inline float embed_prob(const LossArgs& a, int b, long long i, const float* l, float* e_minus_s,
                                   float* v_out) {
    const int z = (int)(i % a.Z);
    const long long t = i / a.Z;
    const int y = (int)(t % a.Y);
    const int x = (int)(t / a.Y);
    const float idx[3] = {(float)x, (float)y, (float)z};
    float ssum = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = tanhf(l[k]);
        const float e = idx[k] + v * a.scale[k];
        const float d = e - a.baked[((long long)b * 3 + k) * a.n + i];
        v_out[k] = v;
        e_minus_s[k] = d;
        ssum += (d * d) * a.inv_var[k];
    }
    return expf(ssum);
}

__global__ void __launch_bounds__(256) loss_reduce_kernel(LossArgs a, int nblk, float* __restrict__ partial) {
    __shared__ float red[4][kLossSums];
    const int b = blockIdx.y, tid = threadIdx.x;
    float s[kLossSums];
#pragma unroll
    for (int k = 0; k < kLossSums; ++k) s[k] = 0.0f;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < a.n; i += (long long)nblk * 256) {
        const float* l = a.logits + ((long long)b * a.n + i) * 5;
        float d[3], v[3];
        const float pe = embed_prob(a, b, i, l, d, v);
        const float ps = sigmoid_(l[3]), pp = sigmoid_(l[4]);
        const float g = a.masks[(long long)b * a.n + i] > 0.0f ? 1.0f : 0.0f;
        const float gs = a.skel[(long long)b * a.n + i] > 0.0f ? 1.0f : 0.0f;
        s[0] += pe * g;
        s[1] += pe * (1.0f - g);
        s[2] += (1.0f - pe) * g;
        s[3] += pp * g;
        s[4] += pp * (1.0f - g);
        s[5] += (1.0f - pp) * g;
        s[6] += ps * gs;
        s[7] += ps * (1.0f - gs);
        s[8] += (1.0f - ps) * gs;
        s[9] += g;
        s[10] += gs;
    }
#pragma unroll
    for (int k = 0; k < kLossSums; ++k) {
        float t = s[k];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) t += __shfl_xor(t, m);
        if ((tid & 63) == 0) red[tid >> 6][k] = t;
    }
    __syncthreads();
    if (tid < kLossSums)
        partial[((long long)b * nblk + blockIdx.x) * kLossSums + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// params (3, 4): alpha, beta, eps, relative weight per loss (embed, prob, skeleton).
// losses (4): embed, prob, skeleton, weighted total (means over the batch, loss.py:155).
// coef (B, 3, 2): d(total)/d(p_i) for a background (g=0) / foreground (g=1) voxel of each loss.
__global__ void __launch_bounds__(256) loss_finalize_kernel(const float* __restrict__ partial, int B, int nblk,
                                                            const float* __restrict__ params,
                                                            float* __restrict__ losses, float* __restrict__ coef) {
    __shared__ double sums[kLossSums];
    __shared__ double acc[256];
    __shared__ double lsum[3];
    const int tid = threadIdx.x;
    if (tid < 3) lsum[tid] = 0.0;
    for (int b = 0; b < B; ++b) {
        for (int k = 0; k < kLossSums; ++k) {
            double s = 0.0;
            for (int j = tid; j < nblk; j += 256) s += (double)partial[((long long)b * nblk + j) * kLossSums + k];
            acc[tid] = s;
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
                for (int j = 0; j < 256; ++j) t += acc[j];
                sums[k] = t;
            }
            __syncthreads();
        }
        if (tid < 3) {
            const double alpha = params[tid * 4], beta = params[tid * 4 + 1], eps = params[tid * 4 + 2];
            const double wgt = (double)params[tid * 4 + 3] / B;
            const double fg = sums[tid == 2 ? 10 : 9];
            double TP = sums[3 * tid], FPs = sums[3 * tid + 1], FNs = sums[3 * tid + 2];
            double c0 = 0.0, c1 = 0.0, loss;
            if (fg > 0.0) {
                const double N = TP + eps;
                const double D = TP + alpha * (FPs + 1e-10) + beta * FNs + eps;
                loss = 1.0 - N / D;
                c0 = wgt * (N * alpha) / (D * D);              // d/dp of a background voxel
                c1 = -wgt * (D - N * (1.0 - beta)) / (D * D);  // d/dp of a foreground voxel
            } else {
                // no instance in the sample: the reference's expanded mask stack is empty, every sum is 0
                loss = 1.0 - eps / (alpha * 1e-10 + eps);
            }
            lsum[tid] += loss;
            coef[((long long)b * 3 + tid) * 2] = (float)c0;
            coef[((long long)b * 3 + tid) * 2 + 1] = (float)c1;
        }
        __syncthreads();
    }
    if (tid == 0) {
        double tot = 0.0;
        for (int k = 0; k < 3; ++k) {
            const double l = lsum[k] / B;
            losses[k] = (float)l;
            tot += (double)params[k * 4 + 3] * l;
        }
        losses[3] = (float)tot;
    }
}

// stand-alone baked_embed_to_prob (lib/embedding_to_prob.py:5-51): planar (B, 3, n) inputs -> (B, n)
__global__ void __launch_bounds__(256) embed_prob_kernel(const float* __restrict__ emb, const float* __restrict__ baked,
                                                         float* __restrict__ out, long long n, float iv0, float iv1,
                                                         float iv2) {
    const int b = blockIdx.y;
    const float iv[3] = {iv0, iv1, iv2};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = emb[((long long)b * 3 + k) * n + i] - baked[((long long)b * 3 + k) * n + i];
            s += (d * d) * iv[k];
        }
        out[(long long)b * n + i] = expf(s);
    }
}

// stand-alone Tversky value (train/loss.py:95-212) of a probability tensor: partial (B, nblk, 4)
__global__ void __launch_bounds__(256) tversky_reduce_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                             long long n, int nblk, float* __restrict__ partial) {
    __shared__ float red[4][4];
    const int b = blockIdx.y, tid = threadIdx.x;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (long long i = (long long)blockIdx.x * 256 + tid; i < n; i += (long long)nblk * 256) {
        const float p = pred[(long long)b * n + i];
        const float g = gt[(long long)b * n + i] != 0.0f ? 1.0f : 0.0f;
        s[0] += p * g;
        s[1] += p * (1.0f - g);
        s[2] += (1.0f - p) * g;
        s[3] += g;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float t = s[k];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) t += __shfl_xor(t, m);
        if ((tid & 63) == 0) red[tid >> 6][k] = t;
    }
    __syncthreads();
    if (tid < 4) partial[((long long)b * nblk + blockIdx.x) * 4 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

__global__ void __launch_bounds__(256) tversky_finalize_kernel(const float* __restrict__ partial, int B, int nblk,
                                                               double alpha, double beta, double eps,
                                                               float* __restrict__ loss) {
    __shared__ double acc[256];
    __shared__ double sums[4];
    const int tid = threadIdx.x;
    double total = 0.0;
    for (int b = 0; b < B; ++b) {
        for (int k = 0; k < 4; ++k) {
            double s = 0.0;
            for (int j = tid; j < nblk; j += 256) s += (double)partial[((long long)b * nblk + j) * 4 + k];
            acc[tid] = s;
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
                for (int j = 0; j < 256; ++j) t += acc[j];
                sums[k] = t;
            }
            __syncthreads();
        }
        if (tid == 0) {
            if (sums[3] > 0.0)
                total += 1.0 - (sums[0] + eps) / (sums[0] + alpha * (sums[1] + 1e-10) + beta * sums[2] + eps);
            else
                total += 1.0 - eps / (alpha * 1e-10 + eps);
        }
        __syncthreads();
    }
    if (tid == 0) loss[0] = (float)(total / B);
}

__global__ void __launch_bounds__(256) loss_bwd_kernel(LossArgs a, const float* __restrict__ coef,
                                                       float* __restrict__ dlogits) {
    const int b = blockIdx.y;
    const float* cf = coef + (long long)b * 6;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
        const float* l = a.logits + ((long long)b * a.n + i) * 5;
        float* o = dlogits + ((long long)b * a.n + i) * 5;
        float d[3], v[3];
        const float pe = embed_prob(a, b, i, l, d, v);
        const float ps = sigmoid_(l[3]), pp = sigmoid_(l[4]);
        const int g = a.masks[(long long)b * a.n + i] > 0.0f ? 1 : 0;
        const int gs = a.skel[(long long)b * a.n + i] > 0.0f ? 1 : 0;
        const float dpe = cf[0 + g];
#pragma unroll
        for (int k = 0; k < 3; ++k)  // d pe/d E_k = pe * 2 (E_k - S_k) * inv_var_k
            o[k] = dpe * pe * (2.0f * d[k] * a.inv_var[k]) * a.scale[k] * (1.0f - v[k] * v[k]);
        o[3] = cf[4 + gs] * ps * (1.0f - ps);
        o[4] = cf[2 + g] * pp * (1.0f - pp);
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient: dW[co][ci][tap] = sum_{b,v} dY[b,v][co] * X[b, in(v, tap)][ci]
// One wave = one 32x32 (cout, cin) tile x one group of taps x one chunk of (b, voxel) pairs,
// reduced on v_mfma_f32_32x32x2_f32 with K = 2 voxels per instruction.
// ------------------------------------------------------------------------------------------
struct WgSrc {
    const float* data;  // (B, xs, ys, zs, C) activated input
    int C, up, Xs, Ys, Zs;
};

struct WgradArgs {
    WgSrc src[2];
    int nsrc;
    const float* dy;  // (B, ox, oy, oz, cout)
    float* part;      // (nchunk, cout, cin, k3)
    float* part_bias; // (nchunk, cout)
    int B, ox, oy, oz, cout, cin, ksize;
    int nchunk, nchunk_b, ncot, ncit, ngroup;
    long long chunk;  // voxels per chunk (even); nchunk = B * nchunk_b
};

template <int NT>  // taps per wave: 9 (k=3, one dx), 8 (k=2), 1 (k=1)
__global__ void __launch_bounds__(64) wgrad_kernel(WgradArgs a) {
    const int lane = threadIdx.x, col = lane & 31, h = lane >> 5;
    int blk = blockIdx.x;
    const int grp = blk % a.ngroup;
    blk /= a.ngroup;
    const int cit = blk % a.ncit;
    blk /= a.ncit;
    const int cot = blk % a.ncot;
    const int chunk = blk / a.ncot;          // chunks never straddle a batch item
    const int b = chunk / a.nchunk_b, cb = chunk % a.nchunk_b;
    const int k = a.ksize, k3 = k * k * k;
    const int stride = (k == 3) ? 1 : k, padw = (k == 3) ? 1 : 0;

    // the cin tile lies inside one source (host checks C % 32 == 0 when there are two)
    int ci0 = 32 * cit, sidx = 0, cbase = 0;
    if (a.nsrc == 2 && ci0 >= a.src[0].C) {
        sidx = 1;
        cbase = a.src[0].C;
    }
    const WgSrc S = a.src[sidx];
    const int ci = ci0 - cbase + col;  // channel inside the source
    const bool ciok = ci < S.C;
    const int co = 32 * cot + col;
    const bool cook = co < a.cout;
    const int Xf = S.up ? S.Xs * 2 : S.Xs, Yf = S.up ? S.Ys * 2 : S.Ys, Zf = S.up ? S.Zs * 2 : S.Zs;
    const long long nvox = (long long)a.ox * a.oy * a.oz;
    const long long svox = (long long)S.Xs * S.Ys * S.Zs;
    // wave-uniform descriptors of this batch item's dy and source (host checks both are < 4 GiB)
    const __amdgpu_buffer_rsrc_t rdy = sk::make_rsrc(a.dy + (long long)b * nvox * a.cout, (unsigned)(nvox * a.cout * 4));
    const __amdgpu_buffer_rsrc_t rsrc = sk::make_rsrc(S.data + (long long)b * svox * S.C, (unsigned)(svox * S.C * 4));

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float bsum = 0.0f;

    const long long q0 = (long long)cb * a.chunk;
    long long q1 = q0 + a.chunk;
    if (q1 > nvox) q1 = nvox;
    // this lane's K slot walks the voxels q0 + h, q0 + h + 2, ...: decode once, then step
    long long q = q0 + h;
    int z = (int)(q % a.oz);
    long long t2 = q / a.oz;
    int y = (int)(t2 % a.oy), x = (int)(t2 / a.oy);
    const long long ntrip = (q1 - q0 + 1) / 2;  // both halves run the same trip count

    // One iteration's operands.  Masked lanes pass an out-of-range offset and get 0 from the buffer bounds
    // check, so the ten loads issue back to back, and the NEXT iteration's loads are issued before this
    // iteration's MFMAs.
    auto fetch = [&](float& av, float (&bv)[NT]) {
        const bool ok = q < q1;
        av = sk::buf_load_f32(rdy, (ok && cook) ? (unsigned)(q * a.cout + co) * 4u : sk::kOob);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int dx = NT == 9 ? grp : (NT == 8 ? (t >> 2) : 0);
            const int dy = NT == 9 ? t / 3 : (NT == 8 ? ((t >> 1) & 1) : 0);
            const int dz = NT == 9 ? t % 3 : (NT == 8 ? (t & 1) : 0);
            int xi = x * stride + dx - padw, yi = y * stride + dy - padw, zi = z * stride + dz - padw;
            const bool inb = xi >= 0 && xi < Xf && yi >= 0 && yi < Yf && zi >= 0 && zi < Zf;
            if (S.up) {
                xi >>= 1;
                yi >>= 1;
                zi >>= 1;
            }
            const unsigned off = ((unsigned)((xi * S.Ys + yi) * S.Zs + zi) * (unsigned)S.C + (unsigned)ci) * 4u;
            bv[t] = sk::buf_load_f32(rsrc, (ok && inb && ciok) ? off : sk::kOob);
        }
        q += 2;
        z += 2;
        while (z >= a.oz) {
            z -= a.oz;
            ++y;
        }
        while (y >= a.oy) {
            y -= a.oy;
            ++x;
        }
    };
    float av_n, bv_n[NT];
    fetch(av_n, bv_n);
    for (long long it = 0; it < ntrip; ++it) {
        const float av = av_n;
        float bv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = bv_n[t];
        fetch(av_n, bv_n);  // past the chunk end every lane is masked and the loads return 0
        bsum += av;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
    }
    // D[row = cout][col = cin]; this lane holds rows (r&3) + 8*(r>>2) + 4*h of column `col`
    float* part = a.part + (long long)chunk * a.cout * a.cin * k3;
    const int cig = ci0 + col;  // channel in the concatenated input
    if (ciok) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tap = grp * NT + t;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * cot + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < a.cout) part[((long long)tap * a.cout + row) * a.cin + cig] = acc[t][r];  // tap-major: lanes (cin) contiguous
            }
        }
    }
    if (a.part_bias && cit == 0 && grp == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (h == 0 && cook) a.part_bias[(long long)chunk * a.cout + co] = bsum;
    }
}

__device__ inline t16 buf_load_f16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(t16, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, byte_off, 0, 0));
}

// Weight gradient of the stem (Cin = 1, k = 3): with one input channel the 27 taps take the place of the input
// channels -- D[cout][tap] += dY[v][cout] * x[v + tap] -- so ONE MFMA per two voxels covers all taps (the generic
// kernel would spend 27 MFMAs with 1 of 32 columns in use).  Partial layout as wgrad_kernel: (chunk, cout, 1, 27).
// DYH: dy is a (scaled) fp16 tensor -- the mixed-precision step; the reduction multiplies by scale[1]
template <bool DYH>
__global__ void __launch_bounds__(64) wgrad_stem_kernel(WgradArgs a) {
    const int lane = threadIdx.x, col = lane & 31, h = lane >> 5;
    int blk = blockIdx.x;
    const int cot = blk % a.ncot;
    const int chunk = blk / a.ncot;
    const int b = chunk / a.nchunk_b, cb = chunk % a.nchunk_b;
    const WgSrc S = a.src[0];
    const int co = 32 * cot + col;
    const bool cook = co < a.cout, tapok = col < 27;
    const int dx = col / 9 - 1, dy = (col / 3) % 3 - 1, dz = col % 3 - 1;  // this lane's tap (B column)
    const long long nvox = (long long)a.ox * a.oy * a.oz;
    constexpr int kDyB = DYH ? 2 : 4;  // bytes per dy element
    const __amdgpu_buffer_rsrc_t rdy = sk::make_rsrc(reinterpret_cast<const char*>(a.dy) + (long long)b * nvox * a.cout * kDyB,
                                                     (unsigned)(nvox * a.cout * kDyB));
    const __amdgpu_buffer_rsrc_t rsrc = sk::make_rsrc(S.data + (long long)b * nvox, (unsigned)(nvox * 4));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float bsum = 0.0f;
    const long long q0 = (long long)cb * a.chunk;
    long long q1 = q0 + a.chunk;
    if (q1 > nvox) q1 = nvox;
    long long q = q0 + h;
    int z = (int)(q % a.oz);
    long long t2 = q / a.oz;
    int y = (int)(t2 % a.oy), x = (int)(t2 / a.oy);
    const long long ntrip = (q1 - q0 + 1) / 2;
    auto fetch = [&](float& av, float& bv) {
        const bool ok = q < q1;
        if constexpr (DYH)
            av = (float)buf_load_f16(rdy, (ok && cook) ? (unsigned)(q * a.cout + co) * 2u : sk::kOob);
        else
            av = sk::buf_load_f32(rdy, (ok && cook) ? (unsigned)(q * a.cout + co) * 4u : sk::kOob);
        const int xi = x + dx, yi = y + dy, zi = z + dz;
        const bool inb = ok && tapok && xi >= 0 && xi < a.ox && yi >= 0 && yi < a.oy && zi >= 0 && zi < a.oz;
        bv = sk::buf_load_f32(rsrc, inb ? (unsigned)((xi * a.oy + yi) * a.oz + zi) * 4u : sk::kOob);
        q += 2;
        z += 2;
        while (z >= a.oz) {
            z -= a.oz;
            ++y;
        }
        while (y >= a.oy) {
            y -= a.oy;
            ++x;
        }
    };
    constexpr int kDepth = 8;   // steps in flight
    float av_n[kDepth], bv_n[kDepth];
#pragma unroll
    for (int u = 0; u < kDepth; ++u) fetch(av_n[u], bv_n[u]);
    for (long long it = 0; it < ntrip; it += kDepth) {
        float av[kDepth], bv[kDepth];
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            av[u] = av_n[u];
            bv[u] = bv_n[u];
        }
#pragma unroll
        for (int u = 0; u < kDepth; ++u) fetch(av_n[u], bv_n[u]);  // past the chunk end: masked, zeros
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            bsum += av[u];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
    }
    float* part = a.part + (long long)chunk * a.cout * 27;
    if (tapok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * cot + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < a.cout) part[(long long)row * 27 + col] = acc[r];
        }
    }
    if (a.part_bias) {
        bsum += __shfl_xor(bsum, 32);
        if (h == 0 && cook) a.part_bias[(long long)chunk * a.cout + co] = bsum;
    }
}

// ------------------------------------------------------------------------------------------
// Mixed-precision pieces (TrainUNet precision="mixed"): fp16 copies of the activations and of the
// (power-of-two scaled) output gradients feed the fast fp16 MFMA kernels; everything else stays fp32.
// ------------------------------------------------------------------------------------------

// |x| maximum of a tensor as float bits in *out (atomicMax on the bit pattern: valid for non-negative floats)
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
    float m = 0.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(out, __float_as_uint(m));
}

// scale[0] = 2^k with max*2^k in [2^12, 2^13) (fp16 headroom for the sums the MFMA forms), scale[1] = 2^-k
__global__ void scale_from_absmax_kernel(const unsigned* __restrict__ mx, float* __restrict__ scale) {
    const float m = __uint_as_float(mx[0]);
    int e = 0;
    if (m > 0.0f && isfinite(m)) {
        frexpf(m, &e);          // m = f * 2^e, f in [0.5, 1)
        e = 13 - e;
    }
    e = e > 60 ? 60 : (e < -60 ? -60 : e);
    scale[0] = ldexpf(1.0f, e);
    scale[1] = ldexpf(1.0f, -e);
}

// 8 elements per lane where the length and the pointers allow (n % 8 == 0: every tensor of the network), else scalar
__global__ void __launch_bounds__(256) cast_f32_f16_kernel(const float* __restrict__ x, t16* __restrict__ y, long long n,
                                                           const float* __restrict__ scale, int vec) {
    const float s = scale ? scale[0] : 1.0f;
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        for (; i < n / 8; i += stride) {
            const float4 a = *reinterpret_cast<const float4*>(x + 8 * i), c = *reinterpret_cast<const float4*>(x + 8 * i + 4);
            half8_t o = {(t16)(a.x * s), (t16)(a.y * s), (t16)(a.z * s), (t16)(a.w * s),
                         (t16)(c.x * s), (t16)(c.y * s), (t16)(c.z * s), (t16)(c.w * s)};
            *reinterpret_cast<half8_t*>(y + 8 * i) = o;
        }
        return;
    }
    for (; i < n; i += stride) y[i] = (t16)(x[i] * s);
}

__global__ void __launch_bounds__(256) cast_f16_f32_kernel(const t16* __restrict__ x, float* __restrict__ y, long long n,
                                                           const float* __restrict__ scale, int accumulate, int vec) {
    const float s = scale ? scale[1] : 1.0f;
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        for (; i < n / 8; i += stride) {
            const half8_t h = *reinterpret_cast<const half8_t*>(x + 8 * i);
            float4 a = {(float)h[0] * s, (float)h[1] * s, (float)h[2] * s, (float)h[3] * s};
            float4 c = {(float)h[4] * s, (float)h[5] * s, (float)h[6] * s, (float)h[7] * s};
            float4* o = reinterpret_cast<float4*>(y + 8 * i);
            if (accumulate) {
                const float4 p0 = o[0], p1 = o[1];
                a = {p0.x + a.x, p0.y + a.y, p0.z + a.z, p0.w + a.w};
                c = {p1.x + c.x, p1.y + c.y, p1.z + c.z, p1.w + c.w};
            }
            o[0] = a;
            o[1] = c;
        }
        return;
    }
    for (; i < n; i += stride) {
        const float v = (float)(x[i]) * s;
        y[i] = accumulate ? y[i] + v : v;
    }
}

// ------------------------------------------------------------------------------------------
// Mixed precision, lean data flow: the block keeps only the RAW fp16 conv output y16; GroupNorm + SiLU writes the
// fp16 activation (plus an fp32 copy only where an fp32 kernel consumes it), and the backward reads y16 and writes
// the fp16 output gradient directly, scaled by a power of two taken from an upper BOUND of its maximum
//   |dy| <= |k0| max|du| + |k1| + max|xh| |k2|   per (sample, channel)
// whose ingredients come out of the reduction pass -- no separate max pass, no fp32 dy, no cast pass.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gn_silu_f16_kernel(const t16* __restrict__ y16, const float* __restrict__ affine,
                                                          t16* __restrict__ z16, float* __restrict__ z32, int C,
                                                          long long n_per_batch) {
    // 8 consecutive channels per lane; the grid stride is a multiple of C/8 vectors: coefficients stay in registers
    const int b = blockIdx.y;
    const long long off = (long long)b * n_per_batch;
    const long long nvec = n_per_batch / 8;
    long long i8 = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c0 = (int)(i8 % (C / 8)) * 8;
    float ga[8], gb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ga[j] = affine[(long long)b * 2 * C + c0 + j];
        gb[j] = affine[(long long)b * 2 * C + C + c0 + j];
    }
    for (; i8 < nvec; i8 += (long long)gridDim.x * 256) {
        const long long i = off + 8 * i8;
        const half8_t yv = *reinterpret_cast<const half8_t*>(y16 + i);
        half8_t o;
        float z[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float u = fmaf(ga[j], (float)yv[j], gb[j]);
            z[j] = u * sigmoid_fast(u);
            o[j] = (t16)z[j];
        }
        *reinterpret_cast<half8_t*>(z16 + i) = o;
        if (z32) {
            *reinterpret_cast<float4*>(z32 + i) = make_float4(z[0], z[1], z[2], z[3]);
            *reinterpret_cast<float4*>(z32 + i + 4) = make_float4(z[4], z[5], z[6], z[7]);
        }
    }
}

// DZH: the incoming gradient is itself a scaled fp16 tensor (the data gradient of the consumer's fast conv, scale
// dz_scale[0]): read 2 bytes and multiply by dz_scale[1] instead of going through an fp32 copy
template <bool DZH>
__device__ inline void load_dz8(const void* __restrict__ dz, long long i, float s, float (&dv)[8]) {
    if constexpr (DZH) {
        const half8_t h = *reinterpret_cast<const half8_t*>(reinterpret_cast<const t16*>(dz) + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) dv[j] = (float)h[j] * s;
    } else {
        const float* p = reinterpret_cast<const float*>(dz) + i;
        const float4 d0 = *reinterpret_cast<const float4*>(p), d1 = *reinterpret_cast<const float4*>(p + 4);
        dv[0] = d0.x; dv[1] = d0.y; dv[2] = d0.z; dv[3] = d0.w;
        dv[4] = d1.x; dv[5] = d1.y; dv[6] = d1.z; dv[7] = d1.w;
    }
}

template <bool DZH>
__global__ void __launch_bounds__(256) gn_bwd_reduce16_kernel(const void* __restrict__ dz, const float* __restrict__ dz_scale,
                                                              const t16* __restrict__ y,
                                                              const float* __restrict__ affine,
                                                              const float* __restrict__ stats, int C, int groups,
                                                              long long voxels, int nblk, float* __restrict__ partial) {
    const float dzs = DZH ? dz_scale[1] : 1.0f;
    // a lane owns 8 consecutive channels of a voxel: one 16-byte load of y, two of dz per voxel (the scalar version --
    // one channel per lane, constants re-read per element -- ran at half of the HBM rate)
    __shared__ float red[256 * 9];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int nq = C / 8, q = tid % nq, row = tid / nq, rows = 256 / nq;
    const int c0 = q * 8, gs = C / groups;
    float a[8], bb[8], mean[8], rstd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = (c0 + j) / gs;
        a[j] = affine[(long long)b * 2 * C + c0 + j];
        bb[j] = affine[(long long)b * 2 * C + C + c0 + j];
        mean[j] = stats[((long long)b * groups + g) * 2];
        rstd[j] = stats[((long long)b * groups + g) * 2 + 1];
    }
    const int vpb = gnb_vox(voxels, C);
    const long long v0 = (long long)blockIdx.x * vpb;
    long long v1 = v0 + vpb;
    if (v1 > voxels) v1 = voxels;
    float s1[8], s2[8], m1[8], m2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = m1[j] = m2[j] = 0.0f;
    for (long long v = v0 + row; v < v1; v += rows) {
        const long long i = ((long long)b * voxels + v) * C + c0;
        const half8_t yv8 = *reinterpret_cast<const half8_t*>(y + i);
        float dv[8];
        load_dz8<DZH>(dz, i, dzs, dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float yv = (float)yv8[j];
            const float du = dv[j] * silu_grad_fast(fmaf(a[j], yv, bb[j]));
            const float xh = (yv - mean[j]) * rstd[j];
            s1[j] += du;
            s2[j] += du * xh;
            m1[j] = fmaxf(m1[j], fabsf(du));
            m2[j] = fmaxf(m2[j], fabsf(xh));
        }
    }
    // block reduction over the rows, one quantity at a time (fixed order: deterministic)
    float* o = partial + ((long long)b * nblk + blockIdx.x) * C * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* src = k == 0 ? s1 : k == 1 ? s2 : k == 2 ? m1 : m2;
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid * 9 + j] = src[j];
        __syncthreads();
        if (tid < C) {
            const int qq = tid / 8, jj = tid % 8;
            float t = 0.0f;
            for (int r = 0; r < rows; ++r) {
                const float v = red[(r * nq + qq) * 9 + jj];
                t = k < 2 ? t + v : fmaxf(t, v);
            }
            o[tid * 4 + k] = t;
        }
        __syncthreads();
    }
}

// one block; as gn_bwd_finalize_kernel plus scale (3 floats): [2^k, 2^-k, bound]
__global__ void __launch_bounds__(1024) gn_bwd_finalize16_kernel(const float* __restrict__ partial, int B, int nblk, int C,
                                                                 int groups, double voxels, const float* __restrict__ gamma,
                                                                 const float* __restrict__ stats, float* __restrict__ coef,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                 float* __restrict__ scale) {
    __shared__ double accv[4096];
    __shared__ double sums[512];  // (C, 4)
    __shared__ double gm[2 * 16];
    __shared__ float bound[128];
    const int tid = threadIdx.x;
    const int nv = 4 * C;
    // a thread owns the four quantities (sum du, sum du*xh, max |du|, max |xh|) of one channel -- one 16-byte load per
    // partial block -- in one of 1024 / C slices of the blocks, eight loads in flight, combined in a fixed order.  (One
    // 4-byte load per thread and iteration kept this one-block kernel at 90 us a call, 1.3 ms of the training step:
    // a single CU with 4 KiB in flight.)
    const int ch = tid % C, sl = tid / C, nsl = 1024 / C;
    double dg = 0.0, db = 0.0;
    float bnd = 0.0f;
    for (int b = 0; b < B; ++b) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        const float4* base = reinterpret_cast<const float4*>(partial + (long long)b * nblk * nv) + ch;
        if (sl < nsl)
            for (int k = sl; k < nblk; k += 8 * nsl) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = k + u * nsl < nblk ? base[(long long)(k + u * nsl) * C] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    s0 += (double)v[u].x;
                    s1 += (double)v[u].y;
                    s2 = (double)v[u].z > s2 ? (double)v[u].z : s2;
                    s3 = (double)v[u].w > s3 ? (double)v[u].w : s3;
                }
            }
        accv[tid * 4 + 0] = s0;
        accv[tid * 4 + 1] = s1;
        accv[tid * 4 + 2] = s2;
        accv[tid * 4 + 3] = s3;
        __syncthreads();
        if (tid < nv) {
            const bool is_max = (tid & 3) >= 2;
            double t = 0.0;
            for (int k = 0; k < nsl; ++k) {
                const double v = accv[(k * C + (tid >> 2)) * 4 + (tid & 3)];
                t = is_max ? (v > t ? v : t) : t + v;
            }
            sums[tid] = t;
        }
        __syncthreads();
        const int gs = C / groups;
        if (tid < groups) {
            double m1 = 0.0, m2 = 0.0;
            for (int c = tid * gs; c < (tid + 1) * gs; ++c) {
                m1 += (double)gamma[c] * sums[4 * c];
                m2 += (double)gamma[c] * sums[4 * c + 1];
            }
            const double n = voxels * gs;
            gm[2 * tid] = m1 / n;
            gm[2 * tid + 1] = m2 / n;
        }
        __syncthreads();
        if (tid < C) {
            const int g = tid / gs;
            const float rstd = stats[((long long)b * groups + g) * 2 + 1];
            float* o = coef + ((long long)b * C + tid) * 3;
            const float k0 = rstd * gamma[tid], k1 = (float)((double)rstd * gm[2 * g]), k2 = (float)((double)rstd * gm[2 * g + 1]);
            o[0] = k0;
            o[1] = k1;
            o[2] = k2;
            db += sums[4 * tid];
            dg += sums[4 * tid + 1];
            bnd = fmaxf(bnd, fabsf(k0) * (float)sums[4 * tid + 2] + fabsf(k1) + (float)sums[4 * tid + 3] * fabsf(k2));
        }
        __syncthreads();
    }
    if (tid < C) {
        dgamma[tid] = (float)dg;
        dbeta[tid] = (float)db;
        bound[tid] = bnd;
    }
    __syncthreads();
    if (tid == 0) {
        float m = 0.0f;
        for (int c = 0; c < C; ++c) m = fmaxf(m, bound[c]);
        int e = 0;
        if (m > 0.0f && isfinite(m)) {
            frexpf(m, &e);
            e = 13 - e;
        }
        e = e > 60 ? 60 : (e < -60 ? -60 : e);
        scale[0] = ldexpf(1.0f, e);
        scale[1] = ldexpf(1.0f, -e);
        scale[2] = m;
    }
}

template <bool DZH>
__global__ void __launch_bounds__(256) gn_bwd_apply16_kernel(const void* __restrict__ dz, const float* __restrict__ dz_scale,
                                                             const t16* __restrict__ y,
                                                             const float* __restrict__ affine,
                                                             const float* __restrict__ stats, const float* __restrict__ coef,
                                                             const float* __restrict__ scale, t16* __restrict__ dy16, int C,
                                                             int groups, long long n_per_batch) {
    // 8 consecutive channels per lane (16-byte y load / dy store, two 16-byte dz loads); the grid stride is a multiple
    // of C/8 vectors, so a lane keeps its channels and their coefficients in registers
    const int b = blockIdx.y;
    const float sc = scale[0];
    const float dzs = DZH ? dz_scale[1] : 1.0f;
    const long long off = (long long)b * n_per_batch;
    const long long nvec = n_per_batch / 8;
    long long i8 = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c0 = (int)(i8 % (C / 8)) * 8, gs = C / groups;
    float a[8], bb[8], mean[8], rstd[8], k0[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c0 + j, g = c / gs;
        a[j] = affine[(long long)b * 2 * C + c];
        bb[j] = affine[(long long)b * 2 * C + C + c];
        mean[j] = stats[((long long)b * groups + g) * 2];
        rstd[j] = stats[((long long)b * groups + g) * 2 + 1];
        const float* k = coef + ((long long)b * C + c) * 3;
        k0[j] = k[0];
        k1[j] = k[1];
        k2[j] = k[2];
    }
    for (; i8 < nvec; i8 += (long long)gridDim.x * 256) {
        const long long i = off + i8 * 8;
        const half8_t yv8 = *reinterpret_cast<const half8_t*>(y + i);
        float dv[8];
        load_dz8<DZH>(dz, i, dzs, dv);
        half8_t out;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float yv = (float)yv8[j];
            const float du = dv[j] * silu_grad_fast(fmaf(a[j], yv, bb[j]));
            out[j] = (t16)((k0[j] * du - k1[j] - ((yv - mean[j]) * rstd[j]) * k2[j]) * sc);
        }
        *reinterpret_cast<half8_t*>(dy16 + i) = out;
    }
}

// Weight gradient on v_mfma_f32_32x32x16_f16: as wgrad_kernel, with K = 16 voxels per instruction.  A lane's
// eight K slots are eight consecutive voxels (lane half h: voxels q + 8h .. q + 8h + 7) of one channel, fetched
// as 16-bit buffer loads (masked lanes read 0 through the bounds check) and packed in registers.

struct Wg16Src {
    const t16* data;  // (B, xs, ys, zs, C) activated input, fp16
    int C, up, Xs, Ys, Zs;
};

struct Wgrad16Args {
    Wg16Src src[2];
    int nsrc;
    const t16* dy;  // (B, ox, oy, oz, cout) fp16, scaled
    float* part;
    float* part_bias;
    int B, ox, oy, oz, cout, cin, ksize;
    int nchunk, nchunk_b, ncot, ncit, ngroup;
    long long chunk;   // voxels per chunk, multiple of 16
    long long* dbg;    // -DSK_TIMING builds: per-wave phase cycle sums of wgrad16x_kernel (sk_debug_set_timing_buffer), else null
};


template <int NT>
__global__ void __launch_bounds__(64) wgrad16_kernel(Wgrad16Args a) {
    const int lane = threadIdx.x, col = lane & 31, h = lane >> 5;
    int blk = blockIdx.x;
    const int grp = blk % a.ngroup;
    blk /= a.ngroup;
    const int cit = blk % a.ncit;
    blk /= a.ncit;
    const int cot = blk % a.ncot;
    const int chunk = blk / a.ncot;
    const int b = chunk / a.nchunk_b, cb = chunk % a.nchunk_b;
    const int k = a.ksize, k3 = k * k * k;
    const int stride = (k == 3) ? 1 : k, padw = (k == 3) ? 1 : 0;
    int ci0 = 32 * cit, sidx = 0, cbase = 0;
    if (a.nsrc == 2 && ci0 >= a.src[0].C) {
        sidx = 1;
        cbase = a.src[0].C;
    }
    const Wg16Src S = a.src[sidx];
    const int ci = ci0 - cbase + col;
    const bool ciok = ci < S.C;
    const int co = 32 * cot + col;
    const bool cook = co < a.cout;
    const int Xf = S.up ? S.Xs * 2 : S.Xs, Yf = S.up ? S.Ys * 2 : S.Ys, Zf = S.up ? S.Zs * 2 : S.Zs;
    const long long nvox = (long long)a.ox * a.oy * a.oz;
    const long long svox = (long long)S.Xs * S.Ys * S.Zs;
    const __amdgpu_buffer_rsrc_t rdy = sk::make_rsrc(a.dy + (long long)b * nvox * a.cout, (unsigned)(nvox * a.cout * 2));
    const __amdgpu_buffer_rsrc_t rsrc = sk::make_rsrc(S.data + (long long)b * svox * S.C, (unsigned)(svox * S.C * 2));

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float bsum = 0.0f;

    const long long q0 = (long long)cb * a.chunk;
    long long q1 = q0 + a.chunk;
    if (q1 > nvox) q1 = nvox;
    const long long ntrip = (q1 - q0 + 15) / 16;

    // operands of the 16 voxels [qb, qb + 16): this lane fetches voxels qb + 8h + j, j = 0..7
    auto fetch = [&](long long qb, half8_t& av, half8_t (&bv)[NT]) {
        long long q = qb + 8 * h;
        int z = (int)(q % a.oz);
        long long t2 = q / a.oz;
        int y = (int)(t2 % a.oy), x = (int)(t2 / a.oy);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = q + j < q1;
            av[j] = buf_load_f16(rdy, (ok && cook) ? (unsigned)((q + j) * a.cout + co) * 2u : sk::kOob);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int dx = NT == 9 ? grp : (NT == 8 ? (t >> 2) : 0);
                const int dy = NT == 9 ? t / 3 : (NT == 8 ? ((t >> 1) & 1) : 0);
                const int dz = NT == 9 ? t % 3 : (NT == 8 ? (t & 1) : 0);
                int xi = x * stride + dx - padw, yi = y * stride + dy - padw, zi = z * stride + dz - padw;
                const bool inb = xi >= 0 && xi < Xf && yi >= 0 && yi < Yf && zi >= 0 && zi < Zf;
                if (S.up) {
                    xi >>= 1;
                    yi >>= 1;
                    zi >>= 1;
                }
                const unsigned off = ((unsigned)((xi * S.Ys + yi) * S.Zs + zi) * (unsigned)S.C + (unsigned)ci) * 2u;
                bv[t][j] = buf_load_f16(rsrc, (ok && inb && ciok) ? off : sk::kOob);
            }
            if (++z == a.oz) {
                z = 0;
                if (++y == a.oy) {
                    y = 0;
                    ++x;
                }
            }
        }
    };
    half8_t av_n, bv_n[NT];
    fetch(q0, av_n, bv_n);
    for (long long it = 0; it < ntrip; ++it) {
        const half8_t av = av_n;
        half8_t bv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = bv_n[t];
        fetch(q0 + 16 * (it + 1), av_n, bv_n);  // past the chunk end every lane is masked: zeros
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum += (float)av[j];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = SK_MFMA_32x32x16_T16(av, bv[t], acc[t], 0, 0, 0);
    }
    float* part = a.part + (long long)chunk * a.cout * a.cin * k3;
    const int cig = ci0 + col;
    if (ciok) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tap = grp * NT + t;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * cot + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < a.cout) part[((long long)tap * a.cout + row) * a.cin + cig] = acc[t][r];  // tap-major: lanes (cin) contiguous
            }
        }
    }
    if (a.part_bias && cit == 0 && grp == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (h == 0 && cook) a.part_bias[(long long)chunk * a.cout + co] = bsum;
    }
}

// The same reduction with whole 16-byte lines: an operand tile of a 16-voxel step is [16 voxels][32 channels]
// fp16 = 1 KiB = ONE LDS-DMA wave instruction (lane = voxel*4 + 8-channel chunk, channels-last lines as they lie
// in HBM), and the K-major fragment the MFMA wants (8 voxels of one channel per lane) comes out of LDS with two
// ds_read_b64_tr_b16 (hardware transpose: per 16-lane group a block of 4 rows x 16 columns, column-major).
// 10 loads per step instead of 80 16-bit loads; tiles double-buffered, next step's DMA in flight under the MFMAs.
typedef SK_TR16_ELEM fp16x4_t __attribute__((__vector_size__(4 * sizeof(SK_TR16_ELEM))));  // the builtin's own vector type

__device__ __forceinline__ void dma16_tile(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ half8_t tr_read_frag(const char* tile, int lane) {
    // lane l: group g = l >> 4 -> channels 16*(g&1) .. +15, voxel half h = g >> 1; lane 4q+p of the group supplies
    // the address of row (voxel) q, columns 4p .. 4p+3 and receives column (l & 15) of the four rows
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const char* base = tile + (8 * (g >> 1) + q) * 64 + (16 * (g & 1) + 4 * p) * 2;
    // (the _v4i16 form of the builtin miscompiles the element extraction on ROCm 7.2: all four lanes of the result
    // read element 0; the _v4f16 form is correct)
    const fp16x4_t lo = SK_DS_READ_TR16_B64((__attribute__((address_space(3))) fp16x4_t*)base);
    const fp16x4_t hi = SK_DS_READ_TR16_B64((__attribute__((address_space(3))) fp16x4_t*)(base + 4 * 64));
    half8_t r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = (t16)lo[j];
        r[4 + j] = (t16)hi[j];
    }
    return r;
}

// Stem weight gradient (Cin = 1, cout = 32) on the 16-bit MFMA: K = 16 voxels of one z row per step where
// wgrad_stem_kernel's fp32 MFMA takes 2 (1.59 ms of the training step for 0.22 ms of MFMA time: one 2-byte and one
// 4-byte load per lane and MFMA).  The dy tile comes by LDS-DMA + transposed reads as in wgrad16t_kernel.  The B operand
// -- column = tap, K = 8 consecutive z of the fp32 image shifted by the tap -- is two 16-byte loads of the aligned
// eight values plus one edge value per lane, then split into a 16-bit value and the 16-bit rounding of its remainder
// (two MFMAs a step): the image keeps 16 (bf16) / 22 (fp16) mantissa bits, so the result differs from the fp32-operand
// kernel by summation order only.  Needs oz % 16 == 0 and whole 16-voxel chunks.
__global__ void __launch_bounds__(64) wgrad_stem16_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) char dyt[2][1024];
    const int lane = threadIdx.x, col = lane & 31, h = lane >> 5;
    const int chunk = blockIdx.x;
    const int b = chunk / a.nchunk_b, cb = chunk % a.nchunk_b;
    const bool tapok = col < 27;
    const int dx = col / 9 - 1, dy = (col / 3) % 3 - 1, dz = col % 3 - 1;  // this lane's tap (B column)
    const long long nvox = (long long)a.ox * a.oy * a.oz;
    const char* dyb = reinterpret_cast<const char*>(a.dy) + (long long)b * nvox * 64 + (lane >> 2) * 64 + (lane & 3) * 16;
    const __amdgpu_buffer_rsrc_t rsrc = sk::make_rsrc(a.src[0].data + (long long)b * nvox, (unsigned)(nvox * 4));
    const long long q0 = (long long)cb * a.chunk;
    long long q1 = q0 + a.chunk;
    if (q1 > nvox) q1 = nvox;
    const int nsteps = q1 > q0 ? (int)((q1 - q0) / 16) : 0;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float bsum = 0.0f;
    int z0 = (int)(q0 % a.oz);
    long long t2 = q0 / a.oz;
    int y = (int)(t2 % a.oy), x = (int)(t2 / a.oy);
    long long q = q0;
    sk::f32x4_t c0, c1;
    float edge;
    auto fetch = [&](int buf) {
        dma16_tile(dyb + q * 64, dyt[buf]);
        const int xi = x + dx, yi = y + dy, z = z0 + 8 * h;
        const bool inb = tapok && xi >= 0 && xi < a.ox && yi >= 0 && yi < a.oy;
        const unsigned row = (unsigned)((xi * a.oy + yi) * a.oz) * 4u;
        c0 = sk::buf_load_f32x4(rsrc, inb ? row + (unsigned)z * 4u : sk::kOob);
        c1 = sk::buf_load_f32x4(rsrc, inb ? row + (unsigned)z * 4u + 16u : sk::kOob);
        const int ze = dz < 0 ? z - 1 : z + 8;   // the value the shifted window takes from outside the aligned eight
        edge = sk::buf_load_f32(rsrc, (inb && dz != 0 && ze >= 0 && ze < a.oz) ? row + (unsigned)ze * 4u : sk::kOob);
        q += 16;
        z0 += 16;
        if (z0 >= a.oz) {
            z0 = 0;
            if (++y >= a.oy) {
                y = 0;
                ++x;
            }
        }
    };
    if (nsteps > 0) fetch(0);
    for (int s = 0; s < nsteps; ++s) {
        const sk::f32x4_t p0 = c0, p1 = c1;
        const float pe = edge;
        if (s + 1 < nsteps) {
            fetch((s + 1) & 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // everything but the four loads just issued
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const half8_t av = tr_read_frag(dyt[s & 1], lane);
        float al[8], f[8];   // the aligned eight; the window shifted by the lane's dz
#pragma unroll
        for (int j = 0; j < 8; ++j) al[j] = j < 4 ? p0[j & 3] : p1[j & 3];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = dz < 0 ? (j == 0 ? pe : al[(j + 7) & 7]) : (dz > 0 ? (j == 7 ? pe : al[(j + 1) & 7]) : al[j]);
        half8_t bh, bl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bh[j] = (t16)f[j];
            bl[j] = (t16)(f[j] - (float)bh[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum += (float)av[j];
        acc = SK_MFMA_32x32x16_T16(av, bh, acc, 0, 0, 0);
        acc = SK_MFMA_32x32x16_T16(av, bl, acc, 0, 0, 0);
    }
    float* part = a.part + (long long)chunk * a.cout * 27;
    if (tapok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            part[(long long)row * 27 + col] = acc[r];
        }
    }
    if (a.part_bias) {
        bsum += __shfl_xor(bsum, 32);
        if (h == 0) a.part_bias[(long long)chunk * a.cout + col] = bsum;
    }
}

template <int NT>
__global__ void __launch_bounds__(64) wgrad16t_kernel(Wgrad16Args a, const char* zero_page) {
    __shared__ __attribute__((aligned(16))) char tiles[2][(NT + 1) * 1024];
    const int lane = threadIdx.x, col = lane & 31, h = lane >> 5;
    const int lv = lane >> 2, lc = lane & 3;  // DMA role: voxel of the step, 8-channel chunk
    int blk = blockIdx.x;
    const int grp = blk % a.ngroup;
    blk /= a.ngroup;
    const int cit = blk % a.ncit;
    blk /= a.ncit;
    const int cot = blk % a.ncot;
    const int chunk = blk / a.ncot;
    const int b = chunk / a.nchunk_b, cb = chunk % a.nchunk_b;
    const int k = a.ksize, k3 = k * k * k;
    const int stride = (k == 3) ? 1 : k, padw = (k == 3) ? 1 : 0;
    int ci0 = 32 * cit, sidx = 0, cbase = 0;
    if (a.nsrc == 2 && ci0 >= a.src[0].C) {
        sidx = 1;
        cbase = a.src[0].C;
    }
    const Wg16Src S = a.src[sidx];
    const int Xf = S.up ? S.Xs * 2 : S.Xs, Yf = S.up ? S.Ys * 2 : S.Ys, Zf = S.up ? S.Zs * 2 : S.Zs;
    const long long nvox = (long long)a.ox * a.oy * a.oz;
    const long long svox = (long long)S.Xs * S.Ys * S.Zs;
    const char* dyb = reinterpret_cast<const char*>(a.dy + (long long)b * nvox * a.cout) + (32 * cot + 8 * lc) * 2;
    const char* sb = reinterpret_cast<const char*>(S.data + (long long)b * svox * S.C) + (ci0 - cbase + 8 * lc) * 2;
    const char* zp = zero_page + lane * 16;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float bsum = 0.0f;

    const long long q0 = (long long)cb * a.chunk;
    long long q1 = q0 + a.chunk;
    if (q1 > nvox) q1 = nvox;
    const int ntrip = q1 > q0 ? (int)((q1 - q0 + 15) / 16) : 0;
    // this lane's DMA voxel walks q0 + lv, q0 + lv + 16, ...
    long long q = q0 + lv;
    int z = (int)(q % a.oz);
    long long t2 = q / a.oz;
    int y = (int)(t2 % a.oy), x = (int)(t2 / a.oy);

    int tdx[NT], tdy[NT], tdz[NT];  // taps of this wave: grp * NT + t
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tap = grp * NT + t;
        tdx[t] = tap / (k * k);
        tdy[t] = (tap / k) % k;
        tdz[t] = tap % k;
    }
    auto issue = [&](int buf) {
        char* tb = tiles[buf];
        const bool ok = q < q1;
        dma16_tile(ok ? dyb + q * a.cout * 2 : zp, tb);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int dx = tdx[t], dy = tdy[t], dz = tdz[t];
            int xi = x * stride + dx - padw, yi = y * stride + dy - padw, zi = z * stride + dz - padw;
            const bool inb = ok && xi >= 0 && xi < Xf && yi >= 0 && yi < Yf && zi >= 0 && zi < Zf;
            if (S.up) {
                xi >>= 1;
                yi >>= 1;
                zi >>= 1;
            }
            dma16_tile(inb ? sb + (long long)((xi * S.Ys + yi) * S.Zs + zi) * S.C * 2 : zp, tb + (t + 1) * 1024);
        }
        q += 16;
        z += 16;
        while (z >= a.oz) {
            z -= a.oz;
            ++y;
        }
        while (y >= a.oy) {
            y -= a.oy;
            ++x;
        }
    };
    if (ntrip > 0) issue(0);
    for (int it = 0; it < ntrip; ++it) {
        const bool more = it + 1 < ntrip;
        if (more) {
            issue((it + 1) & 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT + 1) : "memory");  // this step's tiles have landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const char* tb = tiles[it & 1];
        const half8_t av = tr_read_frag(tb, lane);
        half8_t bv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = tr_read_frag(tb + (t + 1) * 1024, lane);
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum += (float)av[j];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = SK_MFMA_32x32x16_T16(av, bv[t], acc[t], 0, 0, 0);
        // the tile buffer is reused by the DMA issued at the top of the next-but-one iteration; every read of it
        // has completed by then (the MFMAs above consumed them)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float* part = a.part + (long long)chunk * a.cout * a.cin * k3;
    const int cig = ci0 + col;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tap = grp * NT + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * cot + (r & 3) + 8 * (r >> 2) + 4 * h;
            part[((long long)tap * a.cout + row) * a.cin + cig] = acc[t][r];
        }
    }
    if (a.part_bias && cit == 0 && grp == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (h == 0) a.part_bias[(long long)chunk * a.cout + 32 * cot + col] = bsum;
    }
}

// k = 3 and oz % 16 == 0: a step's 16 voxels lie in one z row, so the three dz taps of a (dx, dy) read the SAME 18
// input lines shifted by one -- stage that strip once (18 x 64 B) and take the three K-major fragments from it with
// row-shifted transposed reads.
__device__ __forceinline__ half8_t tr_read_rows(const char* tile, int lane, int row0, int row1) {
    // as tr_read_frag with this lane's two source rows given explicitly (rows of its voxels 8h + q and 8h + 4 + q)
    const int g = lane >> 4, p = lane & 3;
    const char* base = tile + (16 * (g & 1) + 4 * p) * 2;
    const fp16x4_t lo = SK_DS_READ_TR16_B64((__attribute__((address_space(3))) fp16x4_t*)(base + row0 * 64));
    const fp16x4_t hi = SK_DS_READ_TR16_B64((__attribute__((address_space(3))) fp16x4_t*)(base + row1 * 64));
    half8_t r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = (t16)lo[j];
        r[4 + j] = (t16)hi[j];
    }
    return r;
}

__device__ __forceinline__ half8_t tr_read_at(const char* p0, const char* p1) {
    const fp16x4_t lo = SK_DS_READ_TR16_B64((__attribute__((address_space(3))) fp16x4_t*)p0);
    const fp16x4_t hi = SK_DS_READ_TR16_B64((__attribute__((address_space(3))) fp16x4_t*)p1);
    half8_t r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = (t16)lo[j];
        r[4 + j] = (t16)hi[j];
    }
    return r;
}

// Walking y: a wave takes columns (x, 16-voxel z block) and steps through y.  The three rows y-1, y, y+1 of its
// input plane x + dx - 1 sit in a four-slot LDS ring, so a step fetches ONE new strip (row y+2, for the next step)
// and one dy tile: 2.2 KiB per step against 10 for the whole-line kernel above.
#ifndef SK_WG_ABL
#define SK_WG_ABL 0   // experiments only (-DSK_WG_ABL=bits): 1 no tail DMA, 2 no dy DMA, 4 no main DMA, 8 one B read
#endif
__global__ void __launch_bounds__(64, 2) wgrad16y_kernel(Wgrad16Args a, const char* zero_page, int ncol_chunk) {
    constexpr int kSlot = 2048;                                   // strip slot: rows 0..15 | rows 16, 17 (own DMA tile)
    __shared__ __attribute__((aligned(16))) char ring[4 * kSlot];
    __shared__ __attribute__((aligned(16))) char dyt[2][1024];
    const int lane = threadIdx.x, col = lane & 31, h = lane >> 5;
    const int lv = lane >> 2, lc = lane & 3;
    // All waves of a column chunk -- (cout tile, cin tile, dx) -- read the same dy tiles and x planes (one apart for the
    // three dx): block b runs on XCD b % 8 (private L2), so the blocks b, b + 8, b + 16, ... of one XCD, dispatched
    // together, are made the waves of ONE chunk; dealt out in plain order they sat on different XCDs and each fetched its
    // own copy from HBM / Infinity Cache (3.3 TB/s of L2 misses for 22 % MFMA busy).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per = a.ncot * a.ncit * 3;
    int sub = slot % per;
    const int chunk = (slot / per) * 8 + xcd;
    if (chunk >= a.nchunk) return;
    const int grp = sub % 3;  // dx
    sub /= 3;
    const int cit = sub % a.ncit;
    const int cot = sub / a.ncit;
    const int b = chunk / a.nchunk_b, cb = chunk % a.nchunk_b;
    int ci0 = 32 * cit, sidx = 0, cbase = 0;
    if (a.nsrc == 2 && ci0 >= a.src[0].C) {
        sidx = 1;
        cbase = a.src[0].C;
    }
    const Wg16Src S = a.src[sidx];
    const int Xf = S.up ? S.Xs * 2 : S.Xs, Yf = S.up ? S.Ys * 2 : S.Ys;
    const long long nvox = (long long)a.ox * a.oy * a.oz;
    const long long svox = (long long)S.Xs * S.Ys * S.Zs;
    const char* dyb = reinterpret_cast<const char*>(a.dy + (long long)b * nvox * a.cout) + (32 * cot + 8 * lc) * 2;
    const char* sb = reinterpret_cast<const char*>(S.data + (long long)b * svox * S.C) + (ci0 - cbase + 8 * lc) * 2;
    const char* zp = zero_page + lane * 16;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float bsum = 0.0f;
    const bool want_bias = a.part_bias != nullptr && cit == 0 && grp == 0;

    const int nzb = a.oz / 16, ncol = a.ox * nzb;
    const int c0 = cb * ncol_chunk, c1 = min(c0 + ncol_chunk, ncol);
    const int qq = (lane & 15) >> 2;
    const int v0 = 8 * h + qq, v1 = v0 + 4;   // this lane's voxels of the two transposed reads
    const int lane_off = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;   // its 4 columns of a 64-byte tile row

    for (int cidx = c0; cidx < c1; ++cidx) {
        const int x = cidx / nzb, z0 = (cidx % nzb) * 16;
        const int xi = x + grp - 1;
        const bool xok = xi >= 0 && xi < Xf;
        const int xs = S.up ? xi >> 1 : xi;
        const int zs0 = S.up ? ((z0 - 1) >> 1) : (z0 - 1);
        // Column invariants.  Strip of fine row yr -> ring slot (yr + 1) & 3: main tile (rows 0..15) + tail tile (rows
        // 16, 17).  The y loop is unrolled by four (oy % 4 == 0), which makes every ring slot and dy buffer a
        // compile-time constant: the 20 transposed reads of a step take immediate offsets from six per-lane bases and
        // the step's bookkeeping shrinks from ~140 instructions (it, not the MFMAs or the loads, bounded the kernel
        // at one wave per SIMD) to a few pointer increments.
        const int zsm = zs0 + lv, zst = zs0 + 16 + lv;
        const bool mok = xok && zsm >= 0 && zsm < S.Zs, tok = xok && lv < 2 && zst >= 0 && zst < S.Zs;
        const long long rowstride = (long long)S.Zs * S.C * 2;
        const char* colm = sb + ((long long)xs * S.Ys * S.Zs + zsm) * S.C * 2;   // + ys * rowstride
        const char* colt = sb + ((long long)xs * S.Ys * S.Zs + zst) * S.C * 2;
        const long long dystride = (long long)a.oz * a.cout * 2;
        const char* dycol = dyb + ((long long)x * a.oy * a.oz + z0 + lv) * a.cout * 2;      // + y * dystride
        int roff[3][2];   // byte offset inside a slot of this lane's two transposed reads, per dz
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
            const int f0 = z0 + v0 + dz - 1, f1 = z0 + v1 + dz - 1;
            roff[dz][0] = ((S.up ? (f0 >> 1) : f0) - zs0) * 64 + lane_off;
            roff[dz][1] = ((S.up ? (f1 >> 1) : f1) - zs0) * 64 + lane_off;
        }
        auto load_row = [&](int yr, int slot_idx) {
            char* slot = ring + slot_idx * kSlot;
            const bool rowok = yr >= 0 && yr < Yf;
            const long long ro = (long long)(S.up ? yr >> 1 : yr) * rowstride;
            if (!(SK_WG_ABL & 4)) dma16_tile((rowok && mok) ? colm + ro : zp, slot);
            if (!(SK_WG_ABL & 1)) dma16_tile((rowok && tok) ? colt + ro : zp, slot + 1024);
        };
        auto step = [&](int y, auto Uc) {
            constexpr int U = decltype(Uc)::value;   // y & 3
            if (y + 1 < a.oy) {
                load_row(y + 2, (U + 3) & 3);        // overwrites the slot of row y - 2: dead since step y - 1
                if (!(SK_WG_ABL & 2)) dma16_tile(dycol + (long long)(y + 1) * dystride, dyt[(U + 1) & 1]);
                constexpr int kInFlight = 3 - ((SK_WG_ABL & 1) + ((SK_WG_ABL >> 1) & 1) + ((SK_WG_ABL >> 2) & 1));
                if constexpr (kInFlight == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // everything but the three loads just issued
                else if constexpr (kInFlight == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if constexpr (kInFlight == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const half8_t av = tr_read_frag(dyt[U & 1], lane);
            half8_t bv[9];
#pragma unroll
            for (int dyi = 0; dyi < 3; ++dyi) {
                const char* slot = ring + ((U + dyi) & 3) * kSlot;   // row y + dyi - 1
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    if ((SK_WG_ABL & 8) && dyi * 3 + dz > 0) bv[dyi * 3 + dz] = bv[0];
                    else bv[dyi * 3 + dz] = tr_read_at(slot + roff[dz][0], slot + roff[dz][1]);
                }
            }
            if (want_bias) {   // wave-uniform: only the (cin tile 0, dx 0) waves sum dy for the bias gradient
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum += (float)av[j];
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = SK_MFMA_32x32x16_T16(av, bv[t], acc[t], 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        // the previous column's fragments were all read before its last MFMAs were issued: the ring can be refilled
        load_row(-1, 0);
        load_row(0, 1);
        load_row(1, 2);
        dma16_tile(dycol, dyt[0]);
        for (int y = 0; y < a.oy; y += 4) {
            step(y, std::integral_constant<int, 0>{});
            step(y + 1, std::integral_constant<int, 1>{});
            step(y + 2, std::integral_constant<int, 2>{});
            step(y + 3, std::integral_constant<int, 3>{});
        }
    }
    float* part = a.part + (long long)chunk * a.cout * a.cin * 27;
    const int cig = ci0 + col;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tap = grp * 9 + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * cot + (r & 3) + 8 * (r >> 2) + 4 * h;
            part[((long long)tap * a.cout + row) * a.cin + cig] = acc[t][r];
        }
    }
    if (a.part_bias && cit == 0 && grp == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (h == 0) a.part_bias[(long long)chunk * a.cout + 32 * cot + col] = bsum;
    }
}

// Marching along x.  wgrad16y_kernel fetches every input plane three times (the dx = 0, 1, 2 waves of a chunk are
// separate waves, each on its own plane) and has ONE step of LDS-DMA in flight per wave: 22 % MFMA busy, and taking
// the DMA out (MFMAs and LDS reads kept) brings its 10.5 ms of the training step down to 3.1.  Here a workgroup of
// four waves -- one per SIMD, all 27 tap accumulators of a 32 x 32 (cout, cin) tile pair in each -- owns a 16 x 16
// (y, z) footprint and walks x.  The input planes x-1, x, x+1 of the footprint (18 rows: the y halo) sit in a four-slot
// LDS ring and the dy planes in another four-slot ring, so every input plane and every dy plane is fetched ONCE (plus
// halo), 1.6 to 2 x steps -- a step is 3456 MFMA cycles per SIMD -- before its first use; wave w takes output rows
// 4w .. 4w+3.
// The z shift of a tap is taken on dy: with u = z + dz - 1
//     dW[co][ci][dx, dy, dz] = sum dY[x, y, u - dz + 1][co] * X[x + dx - 1, y + dy - 1, u][ci]
// a row step needs three x fragments per plane (rows y-1, y, y+1, unshifted) and three dy fragments (shifts +1, 0, -1):
// 12 fragment reads for 27 MFMAs (15 in the group order used below, which fetches a row's dy fragments twice) where
// the strip kernels make 30 (at full MFMA rate that alone is ~90 % of the LDS bandwidth of a CU).  A source that is a nearest-upsampled half-resolution tensor is expanded by the DMA's own
// per-lane addresses, so the LDS layout is the same for every source.
constexpr int kWxPlane = 18 * 1024;        // input tile of one plane: rows y0-1 .. y0+16, 16 z, 64 B (32 cin) each
constexpr int kWxDy = 16 * 18 * 64;        // dy tile of one plane: 16 rows, z0-1 .. z0+16, 64 B (32 cout) each
constexpr int kWxLds = 4 * kWxPlane + 4 * kWxDy;
// the twelve (row, plane) groups of a step, and which of the two dy fragment register sets a group reads
constexpr int kWxRow[12] = {0, 0, 1, 1, 2, 2, 3, 3, 0, 1, 2, 3};
constexpr int kWxDx[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};
constexpr int kWxSet[12] = {0, 0, 1, 1, 0, 0, 1, 1, 0, 1, 0, 1};
typedef __attribute__((address_space(3))) char* lds_ptr;

#ifndef SK_WX_ABL
#define SK_WX_ABL 0   // timing experiments (-DSK_WX_ABL=bits, results wrong): 1 no LDS-DMA inside the march, 2 no fragment
                      // reads inside the march, 4 the DMA's loads go to (discarded) registers instead of LDS, 8 an
                      // s_waitcnt lgkmcnt(0) right after the step's LDS-DMA burst (timed in slot 7 of the -DSK_TIMING build)
#endif
__device__ __forceinline__ fp16x4_t wx_read(lds_ptr p) {
    return SK_DS_READ_TR16_B64((__attribute__((address_space(3))) fp16x4_t*)p);
}
__device__ __forceinline__ half8_t wx_join(fp16x4_t lo, fp16x4_t hi) {
    half8_t r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = (t16)lo[j];
        r[4 + j] = (t16)hi[j];
    }
    return r;
}
#define SK_WX_PIN() __builtin_amdgcn_sched_barrier(0)
#define SK_WX_RD(p) ((SK_WX_ABL & 2) ? Xl[0] : wx_read(p))

// The 27 accumulators of a wave are 432 registers: more than either register class holds, and the compiler's MFMA
// selection keeps all accumulators in ONE class (it shuttled them through v_accvgpr moves and scratch: 2160 moves, 145
// spilled registers in the loop).  So the MFMA is written out, the first kWxAgprTaps accumulators pinned to AGPRs
// ("+a"), the rest to VGPRs ("+v"): 256 + 176, leaving 80 VGPRs for fragments and addresses.  What the compiler no
// longer knows about these instructions: nothing reads an accumulator before the s_nop block ahead of the epilogue, and a
// tap's MFMAs are nine instructions apart.
// INVARIANT (not checkable by the compiler, pinned by tests/test_hip_train.py::test_conv_wgrad16x_exact_on_integer_operands,
// which runs in the default GPU suite): between two MFMAs on the same accumulator lie the eight MFMAs of the other taps
// of its dx group (SK_WX_M is issued in the fixed order (dy, dz) = (0,0) .. (2,2) per x tap), i.e. >= 8 x 16 pass-cycles
// -- more than the 16-pass latency of v_mfma_f32_32x32x16 --, and nothing reads an accumulator before the s_nop block ahead
// of the epilogue.  Reordering the group bodies or changing kWxAgprTaps must keep both.
constexpr int kWxAgprTaps = 16;
static_assert(kWxAgprTaps >= 0 && kWxAgprTaps <= 27, "27 tap accumulators: 16 in AGPRs + 11 in VGPRs = 432 registers");
#ifdef SK_WX_AGPR_LAST   // experiment: the LAST 16 taps in AGPRs instead of the first 16
#define SK_WX_AGPR_OF(i) ((i) >= 27 - kWxAgprTaps)
#else
#define SK_WX_AGPR_OF(i) ((i) < kWxAgprTaps)
#endif
#ifdef SK_BF16
#define SK_WX_MFMA_ASM "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
#else
#define SK_WX_MFMA_ASM "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0"
#endif
template <bool AGPR>
__device__ __forceinline__ void wx_mfma(f32x16& c, half8_t a, half8_t b) {
    if constexpr (AGPR)
        asm volatile(SK_WX_MFMA_ASM : "+a"(c) : "v"(a), "v"(b));
    else
        asm volatile(SK_WX_MFMA_ASM : "+v"(c) : "v"(a), "v"(b));
}

__global__ void __launch_bounds__(256, 1) wgrad16x_kernel(Wgrad16Args a, int nfy, int nfz, int nxs, int xseg) {
    extern __shared__ __attribute__((aligned(16))) char wx_lds[];
    const lds_ptr L = (lds_ptr)wx_lds;
    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // the (cout tile, cin tile) workgroups of one chunk share its dy / input planes: same XCD, dispatched together
    const int xcd = blockIdx.x & 7, bslot = blockIdx.x >> 3;
    const int per = a.ncot * a.ncit;
    const int sub = bslot % per;
    const int chunk = (bslot / per) * 8 + xcd;
    if (chunk >= a.nchunk) return;
    const int cit = sub % a.ncit, cot = sub / a.ncit;
    int c = chunk;
    const int xsi = c % nxs;
    c /= nxs;
    const int z0 = (c % nfz) * 16;
    c /= nfz;
    const int y0 = (c % nfy) * 16;
    const int b = c / nfy;
    const int xa = xsi * xseg, xb = min(xa + xseg, a.ox);
    int ci0 = 32 * cit, sidx = 0, cbase = 0;
    if (a.nsrc == 2 && ci0 >= a.src[0].C) {
        sidx = 1;
        cbase = a.src[0].C;
    }
    const Wg16Src S = a.src[sidx];
    const int Xf = S.up ? S.Xs * 2 : S.Xs, Yf = S.up ? S.Ys * 2 : S.Ys;
    const long long splane = (long long)S.Ys * S.Zs * S.C * 2, dyplane = (long long)a.oy * a.oz * a.cout * 2;
    const char* sbase = reinterpret_cast<const char*>(S.data) + (long long)b * S.Xs * splane;
    const char* dybase = reinterpret_cast<const char*>(a.dy) + (long long)b * a.ox * dyplane;

    // LDS-DMA roles: an instruction moves 16 positions x 64 B; lane -> position lane >> 2, 16-byte piece lane & 3.
    // Input tile: one instruction per row (wave w: rows w, w + 4, ...); dy tile: 288 positions = 18 instructions
    // (wave w: instructions 3 - w, 7 - w, ...), nine instructions per wave and step.
    const int pos = lane >> 2, c16 = (lane & 3) * 16;
    unsigned xoff[5], doff[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int r = w + 4 * k, y = y0 - 1 + r;
        const bool ok = r < 18 && y >= 0 && y < Yf;
        const int ys = S.up ? y >> 1 : y, zs = S.up ? (z0 + pos) >> 1 : z0 + pos;
        xoff[k] = ok ? (unsigned)((ys * S.Zs + zs) * S.C * 2 + (ci0 - cbase) * 2 + c16) : sk::kOob;
        const int j = 3 - w + 4 * k, p = 16 * j + pos;
        const int pr = p / 18, z = z0 - 1 + p % 18;
        doff[k] = (j < 18 && z >= 0 && z < a.oz) ? (unsigned)(((y0 + pr) * a.oz + z) * a.cout * 2 + 64 * cot + c16) : sk::kOob;
    }
    auto issue_x = [&](int p, int sl) {   // input plane p (fine index; outside the volume: zeros) -> ring slot sl
        const bool pv = p >= 0 && p < Xf;
        const int ps = pv ? (S.up ? p >> 1 : p) : 0;
        const __amdgpu_buffer_rsrc_t rs = sk::make_rsrc(sbase + (long long)ps * splane, (unsigned)splane);
        const lds_ptr dst = L + sl * kWxPlane;
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (w + 4 * k < 18) {
                if (SK_WX_ABL & 4) {   // timing experiment: the same bytes into (discarded) registers instead of LDS
                    sk::f32x4_t junk;
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(junk) : "v"(pv ? xoff[k] : sk::kOob), "s"(rs) : "memory");
                } else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + (w + 4 * k) * 1024), 16,
                                                         pv ? xoff[k] : sk::kOob, 0, 0, 0);
            }
    };
    auto issue_dy = [&](int x, int bi) {
        const __amdgpu_buffer_rsrc_t rs = sk::make_rsrc(dybase + (long long)x * dyplane, (unsigned)dyplane);
        const lds_ptr dst = L + 4 * kWxPlane + bi * kWxDy;
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (3 - w + 4 * k < 18) {
                if (SK_WX_ABL & 4) {
                    sk::f32x4_t junk;
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(junk) : "v"(doff[k]), "s"(rs) : "memory");
                } else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + (3 - w + 4 * k) * 1024), 16,
                                                         doff[k], 0, 0, 0);
            }
    };

    // transposed reads (tr_read_frag): this lane supplies the address of voxel row 8h + q (and + 4), columns 4p .. 4p+3
    // of its 16-channel half
    const int lane_off = (8 * h + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const lds_ptr Lx = L + lane_off + 4 * w * 1024;                       // + slot * kWxPlane + (row + dyi) * 1024
    const lds_ptr Ld = L + 4 * kWxPlane + lane_off + 4 * w * 18 * 64;     // + buf * kWxDy + (row * 18 + 2 - dz) * 64

    f32x16 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float bsum = 0.0f;
    const bool want_bias = a.part_bias != nullptr && cit == 0;
#ifdef SK_TIMING   // per-wave cycle sums: 0 groups 0..6 | 1 landing wait | 2 barrier | 3 groups 7..11 | 4 LDS-DMA issue | 5 prologue | 6 epilogue
    long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev_ = __builtin_readcyclecounter();
#define SK_WX_T(i) { const long long t_ = __builtin_readcyclecounter(); tacc_[i] += t_ - tprev_; tprev_ = t_; }
#else
#define SK_WX_T(i)
#endif
#if defined(SK_TIMING) && defined(SK_WX_TGROUP)   // per-group cycles instead of the phase split: slots 0..11 = groups, 12 issue, 13 wait + barrier
    long long tg_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tgp_ = __builtin_readcyclecounter();
#define SK_WX_TG(i) { const long long t_ = __builtin_readcyclecounter(); tg_[i] += t_ - tgp_; tgp_ = t_; }
#else
#define SK_WX_TG(i)
#endif

    // planes xa-1, xa, xa+1 -> slots 0, 1, 2; dy planes xa, xa+1 -> buffers 0, 1
    issue_x(xa - 1, 0);
    issue_x(xa, 1);
    issue_x(xa + 1, 2);
    issue_dy(xa, 0);
    if (xa + 1 < xb) issue_dy(xa + 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    fp16x4_t Xl[3], Xh[3], Dl[2][3], Dh[2][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {   // fragments of the first group: rows 4w-1 .. 4w+1 of plane xa-1, dy row 4w
        Xl[i] = wx_read(Lx + i * 1024);
        Xh[i] = wx_read(Lx + i * 1024 + 256);
        Dl[0][i] = wx_read(Ld + (2 - i) * 64);
        Dh[0][i] = wx_read(Ld + (2 - i) * 64 + 256);
    }

    // A step = twelve groups of nine MFMAs, (row, plane) in the order of kWxRow / kWxDx: planes x-1 and x for the four
    // rows first, plane x+1 -- the newest -- for the four rows last.  The ONE barrier of a step sits before group 7,
    // whose reads are the first to touch plane x+1: that plane was requested at the start of step x-1, 1.6 steps
    // (~2.3 us at full MFMA rate) earlier, the dy plane two steps earlier.  (With the barrier at the start of the step
    // and one step of lead the layers that miss L2 ran at the loaded HBM latency: 32 -> 32 at 256^3 1170 TFLOP/s, the
    // 64- and 128-channel layers 950, against 1880 for the 96 -> 32 layer, whose tiles mostly hit.)
    // The reads between a group's MFMAs fetch the NEXT group's x fragments (into the registers the group has just
    // finished with) and the dy fragments of the next row (other register set).
    // What is left on the table (tools/bench_wgrad.py, enc0.1 at 256^3, ms per call incl. the partial reduction): 1.00 as
    // is; 0.67 without the LDS-DMA; 0.79 without the fragment reads; 0.62 without both; 0.70 with the same bytes loaded
    // into registers instead of LDS -- neither HBM / L2 nor the LDS reads, the LDS-DMA itself.  Per-group cycle counts
    // (-DSK_TIMING -DSK_WX_TGROUP, `tools/bench_wgrad.py --phases`): the groups 1 .. 11 of a step take 4.5-5.9 % of it
    // each, group 0 -- the first one behind the step's burst of nine buffer_load ... lds per wave -- 19-29 %, the burst's
    // issue 9 %.  An `s_waitcnt lgkmcnt(0)` right behind the burst returns at once (the fragments requested BEFORE it
    // arrive as usual; the DMA does not count there), and which accumulators live in AGPRs does not matter: it is the
    // ds_reads issued BEHIND the burst that stand until its data has landed -- about one L2 / HBM round trip of LDS
    // black-out per burst, for every wave of the CU.  Hence spreading the nine over the groups costs nine round trips
    // (1.60), moving the burst or the barrier changes nothing (0.99), and only loads into registers + ds_write_b128 avoid
    // it: ~36 more VGPRs than the 27 accumulators leave (with the 20 that can be had -- two half-step batches, offsets
    // recomputed per use, three scratch reloads a step -- the writes wait for loads half a step old: 1.25; one load and
    // one write per group with five groups of lead, offsets kept: 256 VGPRs + two in-loop reloads behind vmcnt(0): 1.27).
    SK_WX_T(5)
    for (int x = xa, t = 0; x < xb; ++x, ++t) {
        // slot of plane x-2 (last read before the barrier of step x-1) and buffer of dy plane x-2
        const bool more_x = x + 1 < xb, more_dy = x + 2 < xb;
        SK_WX_T(3)
        if (more_x && !(SK_WX_ABL & 1)) issue_x(x + 2, (t + 3) & 3);
        if (more_dy && !(SK_WX_ABL & 1)) issue_dy(x + 2, (t + 2) & 3);
        SK_WX_T(4)
        SK_WX_TG(12)
        if (SK_WX_ABL & 8) {   // experiment: does an LDS-DMA in flight hold lgkmcnt?  (time of this wait -> slot 7)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SK_WX_T(7)
        }
        lds_ptr xs[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) xs[i] = Lx + ((t + i) & 3) * kWxPlane;   // plane x + i - 1
        const lds_ptr dcur = Ld + (t & 3) * kWxDy, dnext = Ld + ((t + 1) & 3) * kWxDy;
        auto group = [&](auto Gc) {
            constexpr int g = decltype(Gc)::value;
            constexpr int dx = kWxDx[g], cur = kWxSet[g], nxt = cur ^ 1;
            constexpr int gn = (g + 1) % 12;
            // next group's x fragments: plane slot of its dx (the next step's plane x-1 is this step's plane x)
            const lds_ptr xn = (g == 11 ? xs[1] : xs[kWxDx[gn]]) + kWxRow[gn] * 1024;
            // dy fragments of the next row: groups 0 .. 7 fetch half a set each (the row changes every other group),
            // groups 8 .. 11 a whole set
            constexpr int nrow = g < 8 ? (g / 2 + 1) % 4 : (g == 11 ? 0 : g - 7);
            const lds_ptr dn = (g == 11 ? dnext : dcur) + nrow * 18 * 64;
            constexpr int d0 = g < 8 ? 3 * (g & 1) : 0, nd = g < 8 ? 3 : 6;   // reads d0 .. d0 + nd - 1 of the set
            auto dread = [&](auto Ic) {
                constexpr int i = decltype(Ic)::value;   // read i of the set: dz = i / 2, half i % 2
                if constexpr (i >= d0 && i < d0 + nd) {
                    if constexpr (i % 2 == 0)
                        Dl[nxt][i / 2] = SK_WX_RD(dn + (2 - i / 2) * 64);
                    else
                        Dh[nxt][i / 2] = SK_WX_RD(dn + (2 - i / 2) * 64 + 256);
                }
            };
            if (dx == 0 && want_bias) {   // workgroup-uniform: the cin-tile-0 workgroups sum dy for the bias gradient
#pragma unroll
                for (int j = 0; j < 4; ++j) bsum += (float)(t16)Dl[cur][1][j] + (float)(t16)Dh[cur][1][j];
            }
#define SK_WX_M(dyi, dz) \
    wx_mfma<SK_WX_AGPR_OF(dx * 9 + (dyi) * 3 + (dz))>(acc[dx * 9 + (dyi) * 3 + (dz)], wx_join(Dl[cur][dz], Dh[cur][dz]), wx_join(Xl[dyi], Xh[dyi]))
#define SK_WX_D(i) dread(std::integral_constant<int, d0 + (i)>{})
            SK_WX_PIN();
            SK_WX_M(0, 0);
            SK_WX_PIN();
            SK_WX_D(0);
            SK_WX_PIN();
            SK_WX_M(0, 1);
            SK_WX_PIN();
            SK_WX_D(1);
            SK_WX_PIN();
            SK_WX_M(0, 2);
            SK_WX_PIN();
            Xl[0] = SK_WX_RD(xn);
            SK_WX_PIN();
            SK_WX_M(1, 0);
            SK_WX_PIN();
            Xh[0] = SK_WX_RD(xn + 256);
            SK_WX_PIN();
            SK_WX_M(1, 1);
            SK_WX_PIN();
            SK_WX_D(2);
            if constexpr (nd == 6) SK_WX_D(3);
            SK_WX_PIN();
            SK_WX_M(1, 2);
            SK_WX_PIN();
            Xl[1] = SK_WX_RD(xn + 1024);
            SK_WX_PIN();
            SK_WX_M(2, 0);
            SK_WX_PIN();
            Xh[1] = SK_WX_RD(xn + 1024 + 256);
            SK_WX_PIN();
            SK_WX_M(2, 1);
            SK_WX_PIN();
            if constexpr (nd == 6) {
                SK_WX_D(4);
                SK_WX_D(5);
                SK_WX_PIN();
            }
            SK_WX_M(2, 2);
            SK_WX_PIN();
            Xl[2] = SK_WX_RD(xn + 2048);
            Xh[2] = SK_WX_RD(xn + 2048 + 256);
            SK_WX_PIN();
#undef SK_WX_M
#undef SK_WX_D
        };
        group(std::integral_constant<int, 0>{});
        SK_WX_TG(0)
        group(std::integral_constant<int, 1>{});
        SK_WX_TG(1)
        group(std::integral_constant<int, 2>{});
        SK_WX_TG(2)
        group(std::integral_constant<int, 3>{});
        SK_WX_TG(3)
        group(std::integral_constant<int, 4>{});
        SK_WX_TG(4)
        group(std::integral_constant<int, 5>{});
        SK_WX_TG(5)
        group(std::integral_constant<int, 6>{});
        SK_WX_TG(6)
        // everything but this step's own requests has landed (the last steps request less: wait for all)
        SK_WX_T(0)
        if (more_x && more_dy && !(SK_WX_ABL & 1))
            asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SK_WX_T(1)
        __builtin_amdgcn_s_barrier();
        SK_WX_T(2)
        SK_WX_TG(13)
        group(std::integral_constant<int, 7>{});
        SK_WX_TG(7)
        group(std::integral_constant<int, 8>{});
        SK_WX_TG(8)
        group(std::integral_constant<int, 9>{});
        SK_WX_TG(9)
        group(std::integral_constant<int, 10>{});
        SK_WX_TG(10)
        group(std::integral_constant<int, 11>{});
        SK_WX_TG(11)
    }

    SK_WX_T(3)
    // One partial per workgroup: the four waves' accumulators are summed through LDS (fixed order), six taps a round
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results (see wx_mfma)
    __syncthreads();
    sk::f32x4_t* red = reinterpret_cast<sk::f32x4_t*>(wx_lds);   // [tap of the round][wave][r / 4][lane]
    float* part = a.part + (long long)chunk * a.cout * a.cin * 27;
    const int cig = ci0 + col;
#pragma unroll
    for (int t0 = 0; t0 < 27; t0 += 6) {
#pragma unroll
        for (int tt = 0; tt < 6; ++tt) {
            if (t0 + tt < 27) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    sk::f32x4_t v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = acc[t0 + tt][4 * r4 + k];
                    red[((tt * 4 + w) * 4 + r4) * 64 + lane] = v;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 6; ++tt) {
            if (t0 + tt < 27) {
                sk::f32x4_t s = red[((tt * 4 + 0) * 4 + w) * 64 + lane];
#pragma unroll
                for (int ws = 1; ws < 4; ++ws) {
                    const sk::f32x4_t v = red[((tt * 4 + ws) * 4 + w) * 64 + lane];
#pragma unroll
                    for (int k = 0; k < 4; ++k) s[k] += v[k];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {   // accumulator register r = 4w + k of the MFMA tile
                    const int row = 32 * cot + k + 8 * w + 4 * h;
                    part[((long long)(t0 + tt) * a.cout + row) * a.cin + cig] = s[k];
                }
            }
        }
        __syncthreads();
    }
    if (want_bias) {
        float* rb = reinterpret_cast<float*>(wx_lds);
        bsum += __shfl_xor(bsum, 32);
        if (h == 0) rb[w * 32 + col] = bsum;
        __syncthreads();
        if (w == 0 && h == 0) a.part_bias[(long long)chunk * a.cout + 32 * cot + col] = (rb[col] + rb[32 + col]) + (rb[64 + col] + rb[96 + col]);
    }
#ifdef SK_TIMING
    SK_WX_T(6)
    if (a.dbg && lane == 0 && blockIdx.x < (unsigned)sk::kTimingBlocks)
#ifdef SK_WX_TGROUP
        for (int i = 0; i < 16; ++i) a.dbg[((long long)blockIdx.x * 4 + w) * sk::kTimingSlots + i] = tg_[i];
#else
        for (int i = 0; i < 8; ++i) a.dbg[((long long)blockIdx.x * 4 + w) * sk::kTimingSlots + i] = tacc_[i];
#endif
#endif
#undef SK_WX_TG
#undef SK_WX_T
}

// Device-side counterpart of sk_conv3d_pack_weight_host (weights change every step in training): fp32
// torch-layout weight -> fp16 MFMA A fragments.  `transposed`: pack the data-gradient operator instead,
// W'[co'][ci'][tap] = W[ci'][c_lo + co'][flipped tap]  (co' < eff_cout input channels of the layer, ci' <
// eff_cin = the layer's output channels).  One thread per fp16 element.
struct PackArgs {
    const float* w;  // (Co, Ci, k, k, k)
    t16* dst;
    int Co, Ci, ksize, eff_cout, eff_cin, transposed, c_lo;
    long long n;     // halves to write
};

__global__ void __launch_bounds__(256) pack_weight_kernel(PackArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
    long long f = i >> 9;  // fragment index
    const int NT = a.eff_cout / 32, k = a.ksize, k3 = k * k * k;
    int nt, c0, kx, ky, kz;
    int co, ci;
    if (k == 3 && a.eff_cout == 32) {  // 16x16x32 fragments (the COUT-32 conv kernel): [chunk32][dy*3+dz][i(2)][dx],
        nt = 0;                        // lane: cout 16i + (l&15), cin 32ch + 8(l>>4) + j
        const int dx = (int)(f % 3);
        f /= 3;
        const int ih = (int)(f % 2);
        f /= 2;
        const int dydz = (int)(f % 9);
        const int ch = (int)(f / 9);
        c0 = ch * 32;
        kx = dx;
        ky = dydz / 3;
        kz = dydz % 3;
        co = 16 * ih + (l & 15);
        ci = c0 + 8 * (l >> 4) + j;
    } else if (k == 3) {               // 32x32x16 fragments: [chunk32][dy*3+dz][ks][dx][nt]
        nt = (int)(f % NT);
        f /= NT;
        const int dx = (int)(f % 3);
        f /= 3;
        const int ks = (int)(f % 2);
        f /= 2;
        const int dydz = (int)(f % 9);
        const int ch = (int)(f / 9);
        c0 = ch * 32 + ks * 16;
        kx = dx;
        ky = dydz / 3;
        kz = dydz % 3;
        co = 32 * nt + (l & 31);
        ci = c0 + 8 * (l >> 5) + j;
    } else {       // [tap][ks][nt]
        nt = (int)(f % NT);
        f /= NT;
        const int nks = a.eff_cin / 16;
        const int ks = (int)(f % nks);
        const int tap = (int)(f / nks);
        c0 = ks * 16;
        kx = tap / (k * k);
        ky = (tap / k) % k;
        kz = tap % k;
        co = 32 * nt + (l & 31);
        ci = c0 + 8 * (l >> 5) + j;
    }
    int tap = (kx * k + ky) * k + kz;
    long long src;
    if (a.transposed) {
        if (a.transposed == 1) tap = k3 - 1 - tap;  // 2: transposed without the flip (stride-2 scatter form)
        src = ((long long)ci * a.Ci + a.c_lo + co) * k3 + tap;
    } else {
        src = ((long long)co * a.Ci + ci) * k3 + tap;
    }
    a.dst[i] = (t16)(a.w[src]);
}

// Data gradient of a k = 2, stride-2 conv from its eight per-parity pointwise products: t16 (8, B, cx, cy, cz, C)
// holds T_p[c] = W_p^T dY[c]; dX[2c + p] (+)= T_p[c] * scale[1].
// add16 (optional): another scaled fp16 gradient of the same fine tensor (the decoder's contribution to a skip tensor,
// scale add_scale), summed in here instead of through an fp32 copy + accumulate pass
__global__ void __launch_bounds__(256) interleave2_kernel(const t16* __restrict__ tens16, float* __restrict__ dx, int B,
                                                          int cx, int cy, int cz, int C, const float* __restrict__ scale,
                                                          int accumulate, const t16* __restrict__ add16,
                                                          const float* __restrict__ add_scale) {
    // 8 channels per lane: one 16-byte load of the parity tensor, two 16-byte stores (+ loads when accumulating)
    const float s = scale ? scale[1] : 1.0f;
    const int nq = C / 8;
    const long long ncoarse = (long long)B * cx * cy * cz * C;
    const long long n = (long long)B * cx * cy * cz * 8 * nq;   // fine voxels x channel octets
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int q = (int)(i % nq);
        long long t = i / nq;
        const int z = (int)(t % (2 * cz));
        t /= 2 * cz;
        const int y = (int)(t % (2 * cy));
        t /= 2 * cy;
        const int x = (int)(t % (2 * cx));
        const int b = (int)(t / (2 * cx));
        const int p = ((x & 1) << 2) | ((y & 1) << 1) | (z & 1);
        const long long ci = ((((long long)b * cx + (x >> 1)) * cy + (y >> 1)) * cz + (z >> 1)) * C + 8 * q;
        const half8_t h = *reinterpret_cast<const half8_t*>(tens16 + (long long)p * ncoarse + ci);
        float4 v0 = {(float)h[0] * s, (float)h[1] * s, (float)h[2] * s, (float)h[3] * s};
        float4 v1 = {(float)h[4] * s, (float)h[5] * s, (float)h[6] * s, (float)h[7] * s};
        float4* o = reinterpret_cast<float4*>(dx + i * 8);
        if (add16) {
            const float as = add_scale[1];
            const half8_t g = *reinterpret_cast<const half8_t*>(add16 + i * 8);
            v0 = {fmaf((float)g[0], as, v0.x), fmaf((float)g[1], as, v0.y), fmaf((float)g[2], as, v0.z), fmaf((float)g[3], as, v0.w)};
            v1 = {fmaf((float)g[4], as, v1.x), fmaf((float)g[5], as, v1.y), fmaf((float)g[6], as, v1.z), fmaf((float)g[7], as, v1.w)};
        }
        if (accumulate) {
            const float4 p0 = o[0], p1 = o[1];
            v0 = {p0.x + v0.x, p0.y + v0.y, p0.z + v0.z, p0.w + v0.w};
            v1 = {p1.x + v1.x, p1.y + v1.y, p1.z + v1.z, p1.w + v1.w};
        }
        o[0] = v0;
        o[1] = v1;
    }
}

// ------------------------------------------------------------------------------------------
// 16-bit hand-off of the gradients that used to leave their producer as fp32 (the skip tensors' stride-2 interleave, the
// 2x2x2 sum pooling under an upsampled source, the heads' data gradient): the consumer is a GroupNorm backward that
// reads a scaled 16-bit dz as it is (gn_bwd_*16_kernel<true>), so the fp32 tensor cost 4 B/element written once and
// read twice.  The output scale is a power of two derived from the INPUT scales, so that the result cannot overflow:
// out_scale (3 floats, written by the kernel) = [s, 1/s, bound on |dx| or NaN].  Slot [2] is a bound only where the
// producer knows one (gn_bwd_*16: the GroupNorm-backward reduction yields it; heads_dgrad_h_kernel: 2^13 / s by
// construction).  The interleave and the pooling below only know the absmax of the dy their INPUT was derived from --
// W^T dy can exceed it by sum |W| --, so they write NaN ("unknown"): a consumer that trusted a number there would
// mis-scale silently, a NaN it cannot miss.  Today's consumers (gn_bwd_*16_kernel<true>) read [1] only.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) interleave2_h_kernel(const t16* __restrict__ tens16, t16* __restrict__ out16, int B, int cx,
                                                            int cy, int cz, int C, const float* __restrict__ scale,
                                                            const t16* __restrict__ add16, const float* __restrict__ add_scale,
                                                            float* __restrict__ out_scale) {
    // dx * s_out = T16 * (s_out / s_T) + add16 * (s_out / s_add): with s_out = min(s_T, s_add) / 2 both factors are <= 1/2,
    // the sum stays below the larger 16-bit operand's range; without a second operand the parity tensor passes through
    const float sT = scale[0];
    const float s_out = add16 ? fminf(sT, add_scale[0]) * 0.5f : sT;
    const float fa = s_out * scale[1], fb = add16 ? s_out * add_scale[1] : 0.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out_scale[0] = s_out;
        out_scale[1] = 1.0f / s_out;
        out_scale[2] = __builtin_nanf("");   // unknown: see above
    }
    const int nq = C / 8;
    const long long ncoarse = (long long)B * cx * cy * cz * C;
    const long long n = (long long)B * cx * cy * cz * 8 * nq;   // fine voxels x channel octets
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int q = (int)(i % nq);
        long long t = i / nq;
        const int z = (int)(t % (2 * cz));
        t /= 2 * cz;
        const int y = (int)(t % (2 * cy));
        t /= 2 * cy;
        const int x = (int)(t % (2 * cx));
        const int b = (int)(t / (2 * cx));
        const int p = ((x & 1) << 2) | ((y & 1) << 1) | (z & 1);
        const long long ci = ((((long long)b * cx + (x >> 1)) * cy + (y >> 1)) * cz + (z >> 1)) * C + 8 * q;
        const half8_t h = *reinterpret_cast<const half8_t*>(tens16 + (long long)p * ncoarse + ci);
        half8_t o;
        if (add16) {
            const half8_t g = *reinterpret_cast<const half8_t*>(add16 + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (t16)fmaf((float)g[j], fb, (float)h[j] * fa);
        } else {
            o = h;
        }
        *reinterpret_cast<half8_t*>(out16 + i * 8) = o;
    }
}

// coarse * (s / 8) = (sum of the 8 children, each scaled by s) / 8
__global__ void __launch_bounds__(256) sumpool2_hh_kernel(const t16* __restrict__ fine, const float* __restrict__ scale,
                                                          t16* __restrict__ coarse, float* __restrict__ out_scale, int B, int cx,
                                                          int cy, int cz, int C) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out_scale[0] = scale[0] * 0.125f;
        out_scale[1] = scale[1] * 8.0f;
        out_scale[2] = __builtin_nanf("");   // unknown: see interleave2_h_kernel
    }
    const int nq = C / 8;
    const long long n = (long long)B * cx * cy * cz * nq;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int q = (int)(i % nq);
        long long t = i / nq;
        const int z = (int)(t % cz);
        t /= cz;
        const int y = (int)(t % cy);
        t /= cy;
        const int x = (int)(t % cx);
        const int b = (int)(t / cx);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const long long fi = (((long long)b * 2 * cx + 2 * x + (d >> 2)) * 2 * cy + 2 * y + ((d >> 1) & 1)) * 2 * cz +
                                 2 * z + (d & 1);
            const half8_t h = *reinterpret_cast<const half8_t*>(fine + fi * C + 8 * q);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)h[j];
        }
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (t16)(acc[j] * 0.125f);
        *reinterpret_cast<half8_t*>(coarse + i * 8) = o;
    }
}

// Data gradient of the heads (K = 5 logits -> C features) as a scaled 16-bit tensor: dx[v][c] = sum_k dl[v][k] W[k][c].
// dl_scale = sk_train_absmax_scale(dl): max |dl| * dl_scale[0] in [2^12, 2^13); with 2^e >= max_c sum_k |W[k][c]| the scale
// s = dl_scale[0] * 2^-e keeps |dx| * s below 2^13.  8 channels per lane (one 16-byte store), the voxel's lanes share dl.
template <int K>
__global__ void __launch_bounds__(256) heads_dgrad_h_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                                            t16* __restrict__ dx16, float* __restrict__ out_scale,
                                                            const float* __restrict__ dl_scale, long long nvox, int C) {
    float wmax = 0.0f;
    for (int c = 0; c < C; ++c) {
        float sc = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) sc += fabsf(w[k * C + c]);
        wmax = fmaxf(wmax, sc);
    }
    int e = 0;
    if (wmax > 0.0f && isfinite(wmax)) frexpf(wmax, &e);   // wmax < 2^e
    e = e > 60 ? 60 : (e < -60 ? -60 : e);
    const float s = dl_scale[0] * ldexpf(1.0f, -e);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out_scale[0] = s;
        out_scale[1] = 1.0f / s;
        out_scale[2] = ldexpf(1.0f, 13) / s;
    }
    const int nq = C / 8;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int q = (int)(i % nq);   // fixed per lane: 256 and the grid stride are multiples of nq
    float wk[K][8];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) wk[k][j] = w[k * C + 8 * q + j] * s;
    for (; i < nvox * nq; i += (long long)gridDim.x * 256) {
        const long long v = i / nq;
        float g[K];
#pragma unroll
        for (int k = 0; k < K; ++k) g[k] = dl[v * K + k];
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float r = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) r = fmaf(g[k], wk[k][j], r);
            o[j] = (t16)r;
        }
        *reinterpret_cast<half8_t*>(dx16 + v * C + 8 * q) = o;
    }
}

// out = sum over chunks of part[c] (fixed order -> deterministic): 64 elements per block, four chunk slices.
// k3 > 0: the partials are tap-major (tap, cout, cin) -- written with the lanes (cin) contiguous -- and the result
// goes to the torch layout (cout, cin, k3); k3 == 0: same layout in and out.
constexpr int kWredSlices = 16;  // chunk slices per block (x 64 elements = 1024 threads)
__global__ void __launch_bounds__(64 * kWredSlices) wgrad_reduce_kernel(const float* __restrict__ part, int nchunk, long long n,
                                                                        float* __restrict__ out, const float* __restrict__ scale,
                                                                        int cout, int cin, int k3) {
    __shared__ double red[64 * kWredSlices];
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + e;
    // four independent loads in flight per lane (a single dependent chain over nchunk / 4 strided loads made the
    // 30 launches of a step latency-bound: 82 us each)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i < n) {
        int c = sl;
        for (; c + 3 * kWredSlices < nchunk; c += 4 * kWredSlices) {
            const float v0 = part[(long long)c * n + i], v1 = part[(long long)(c + kWredSlices) * n + i];
            const float v2 = part[(long long)(c + 2 * kWredSlices) * n + i], v3 = part[(long long)(c + 3 * kWredSlices) * n + i];
            s0 += (double)v0;
            s1 += (double)v1;
            s2 += (double)v2;
            s3 += (double)v3;
        }
        for (; c < nchunk; c += kWredSlices) s0 += (double)part[(long long)c * n + i];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && i < n) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < kWredSlices; ++k) s += red[64 * k + e];
        long long o = i;
        if (k3 > 0) {
            const int ci = (int)(i % cin);
            const long long t = i / cin;
            const int row = (int)(t % cout), tap = (int)(t / cout);
            o = ((long long)row * cin + ci) * k3 + tap;
        }
        out[o] = scale ? (float)(s * (double)scale[1]) : (float)s;
    }
}

// ------------------------------------------------------------------------------------------
// 2x2x2 sum pooling: backward of the nearest x2 upsampling folded into the decoder convs
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sumpool2_kernel(const float* __restrict__ fine, float* __restrict__ coarse,
                                                       int B, int cx, int cy, int cz, int C) {
    const long long n = (long long)B * cx * cy * cz * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int z = (int)(t % cz);
        t /= cz;
        const int y = (int)(t % cy);
        t /= cy;
        const int x = (int)(t % cx);
        const int b = (int)(t / cx);
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const long long fi = (((long long)b * 2 * cx + 2 * x + (d >> 2)) * 2 * cy + 2 * y + ((d >> 1) & 1)) * 2 * cz +
                                 2 * z + (d & 1);
            s += fine[fi * C + c];
        }
        coarse[i] = s;
    }
}

// the same from a scaled fp16 fine tensor (the fast data-gradient conv's output), 8 channels per lane: no fp32 copy of
// the fine gradient is ever written
__global__ void __launch_bounds__(256) sumpool2_f16_kernel(const t16* __restrict__ fine, const float* __restrict__ scale,
                                                           float* __restrict__ coarse, int B, int cx, int cy, int cz, int C) {
    const float s = scale[1];
    const int nq = C / 8;
    const long long n = (long long)B * cx * cy * cz * nq;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int q = (int)(i % nq);
        long long t = i / nq;
        const int z = (int)(t % cz);
        t /= cz;
        const int y = (int)(t % cy);
        t /= cy;
        const int x = (int)(t % cx);
        const int b = (int)(t / cx);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const long long fi = (((long long)b * 2 * cx + 2 * x + (d >> 2)) * 2 * cy + 2 * y + ((d >> 1) & 1)) * 2 * cz +
                                 2 * z + (d & 1);
            const half8_t h = *reinterpret_cast<const half8_t*>(fine + fi * C + 8 * q);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)h[j];
        }
        float* o = coarse + i * 8;
        *reinterpret_cast<float4*>(o) = make_float4(acc[0] * s, acc[1] * s, acc[2] * s, acc[3] * s);
        *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4] * s, acc[5] * s, acc[6] * s, acc[7] * s);
    }
}

// ------------------------------------------------------------------------------------------
// Heads (1x1x1 conv, 32 features -> 5 logits) of the mixed step, straight on the fp16 activation: both are pure
// HBM streams (64 B in, 20 B out per voxel), so no matrix instruction.  Four lanes share a voxel (8 channels = 16 bytes
// each), partial dot products are combined with two xor-shuffles.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) heads_fwd_f16_kernel(const t16* __restrict__ z, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ logits,
                                                            long long nvox) {
    const int q = threadIdx.x & 3;
    float wk[5][8];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) wk[k][j] = w[k * 32 + 8 * q + j];
    const long long stride = (long long)gridDim.x * 64;
    for (long long v = (long long)blockIdx.x * 64 + (threadIdx.x >> 2); v < nvox; v += stride) {
        const half8_t h = *reinterpret_cast<const half8_t*>(z + v * 32 + 8 * q);
        float acc[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = (float)h[j];
#pragma unroll
            for (int k = 0; k < 5; ++k) acc[k] = fmaf(x, wk[k][j], acc[k]);
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            acc[k] += __shfl_xor(acc[k], 1);
            acc[k] += __shfl_xor(acc[k], 2);
        }
        // lane q of the voxel writes logit q; lane 0 also logit 4
        float* o = logits + v * 5;
        const float mine = q == 0 ? acc[0] : q == 1 ? acc[1] : q == 2 ? acc[2] : acc[3];
        o[q] = mine + bias[q];
        if (q == 0) o[4] = acc[4] + bias[4];
    }
}

// dW[k][c] = sum_v dl[v][k] z[v][c], db[k] = sum_v dl[v][k]: per block partials (fixed order), reduced by wgrad_reduce_kernel
constexpr int kHeadsVox = 8192;  // voxels per block
__global__ void __launch_bounds__(256) heads_wgrad_f16_kernel(const t16* __restrict__ z, const float* __restrict__ dl,
                                                              float* __restrict__ part, float* __restrict__ part_b,
                                                              long long nvox) {
    __shared__ float red[64][41];   // 41: odd pitch, conflict-free column sums
    const int q = threadIdx.x & 3, row = threadIdx.x >> 2;
    const long long v0 = (long long)blockIdx.x * kHeadsVox;
    long long v1 = v0 + kHeadsVox;
    if (v1 > nvox) v1 = nvox;
    float acc[5][8];
    float accb[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.0f;
    for (long long v = v0 + row; v < v1; v += 64) {
        const half8_t h = *reinterpret_cast<const half8_t*>(z + v * 32 + 8 * q);
        float g[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) g[k] = dl[v * 5 + k];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            accb[k] += g[k];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[k][j] = fmaf(g[k], (float)h[j], acc[k][j]);
        }
    }
    // reduce over the 64 rows, one lane quarter (8 channels) per pass, through LDS in a fixed order
    for (int pass = 0; pass < 4; ++pass) {
        __syncthreads();
        if (q == pass) {
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) red[row][k * 8 + j] = acc[k][j];
        }
        __syncthreads();
        if (threadIdx.x < 40) {
            float t = 0.0f;
            for (int r = 0; r < 64; ++r) t += red[r][threadIdx.x];
            const int k = threadIdx.x / 8, j = threadIdx.x % 8;
            part[(long long)blockIdx.x * 160 + k * 32 + 8 * pass + j] = t;
        }
    }
    __syncthreads();
    if (q == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) red[row][k] = accb[k];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        float t = 0.0f;
        for (int r = 0; r < 64; ++r) t += red[r][threadIdx.x];
        part_b[(long long)blockIdx.x * 5 + threadIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------
// AdamW (torch.optim.AdamW semantics: decoupled decay, bias-corrected moments)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long long n,
                                                    float lr, float beta1, float beta2, float eps, float wd,
                                                    float bc1, float bc2_sqrt) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i];
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

int fill_loss_args(LossArgs& a, const float* logits, const float* masks, const float* skel, const float* baked,
                   int X, int Y, int Z, const float* scale_host, const float* sigma_host) {
    a.logits = logits;
    a.masks = masks;
    a.skel = skel;
    a.baked = baked;
    a.n = (long long)X * Y * Z;
    a.Y = Y;
    a.Z = Z;
    for (int k = 0; k < 3; ++k) {
        a.scale[k] = scale_host[k];
        const float s = sigma_host[k] + 1e-16f;  // embedding_to_prob.py:38-39, fp32 arithmetic
        a.inv_var[k] = 1.0f / (s * s * 2.0f * -1.0f);
    }
    return 0;
}

}  // namespace

extern "C" {

int sk_train_gn_silu(const float* y, const float* affine, float* z, int B, int64_t voxels, int C, void* stream) {
    SK_CHECK_ARG(y && affine && z && C > 0, "sk_train_gn_silu: bad arguments");
    long long n = voxels * C;
    dim3 grid(sk::stream_grid(n, 256, 4), B);
    gn_silu_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(y, affine, z, C, n);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_train_gn_bwd_num_blocks(int64_t voxels) {  // upper bound over the channel counts (workspace sizing)
    const int vpb = gnb_vox(voxels, 32);
    return (int)((voxels + vpb - 1) / vpb);
}

int sk_train_gn_silu_bwd(const float* dz, const float* y, const float* affine, const float* stats,
                         const float* gamma, int B, int64_t voxels, int C, int groups, float* dy, float* dgamma,
                         float* dbeta, float* workspace, void* stream) {
    SK_CHECK_ARG(dz && y && affine && stats && gamma && dy && dgamma && dbeta && workspace,
                 "sk_train_gn_silu_bwd: NULL pointer");
    SK_CHECK_ARG((C == 32 || C == 64 || C == 128) && groups > 0 && groups <= 16 && C % groups == 0,
                 "sk_train_gn_silu_bwd: C=%d groups=%d unsupported", C, groups);
    const int nblk = gnb_blocks(voxels, C);                  // <= sk_train_gn_bwd_num_blocks: the workspace fits
    float* partial = workspace;                              // (B, nblk, C, 2)
    float* coef = workspace + (long long)B * nblk * C * 2;   // (B, C, 3)
    gn_bwd_reduce_kernel<<<dim3(nblk, B), 256, 0, (hipStream_t)stream>>>(dz, y, affine, stats, C, groups, voxels, nblk,
                                                                         partial);
    SK_CHECK_LAUNCH();
    gn_bwd_finalize_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(partial, B, nblk, C, groups, (double)voxels, gamma,
                                                                stats, coef, dgamma, dbeta);
    SK_CHECK_LAUNCH();
    long long n = voxels * C;
    gn_bwd_apply_kernel<<<dim3(sk::stream_grid(n, 256, 4), B), 256, 0, (hipStream_t)stream>>>(dz, y, affine, stats, coef,
                                                                                               dy, C, groups, n);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int64_t sk_train_gn_bwd_workspace_floats(int B, int64_t voxels, int C) {
    return (int64_t)B * sk_train_gn_bwd_num_blocks(voxels) * C * 2 + (int64_t)B * C * 3;
}

int sk_train_loss_num_blocks(int64_t voxels) {
    int64_t g = (voxels + 256 * 16 - 1) / (256 * 16);
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

int64_t sk_train_loss_workspace_floats(int B, int64_t voxels) {
    return (int64_t)B * sk_train_loss_num_blocks(voxels) * kLossSums + (int64_t)B * 6;
}

int sk_train_loss(const float* logits, const float* masks, const float* skeleton_masks, const float* baked, int B,
                  int X, int Y, int Z, const float* vector_scale_host, const float* sigma_host,
                  const float* loss_params_host, float* losses, float* dlogits, float* workspace, void* stream) {
    SK_CHECK_ARG(logits && masks && skeleton_masks && baked && losses && workspace, "sk_train_loss: NULL pointer");
    SK_CHECK_ARG(vector_scale_host && sigma_host && loss_params_host, "sk_train_loss: NULL host parameter");
    SK_CHECK_ARG(B >= 1 && X >= 1 && Y >= 1 && Z >= 1, "sk_train_loss: bad extents");
    LossArgs a{};
    fill_loss_args(a, logits, masks, skeleton_masks, baked, X, Y, Z, vector_scale_host, sigma_host);
    const int nblk = sk_train_loss_num_blocks(a.n);
    float* partial = workspace;
    float* coef = workspace + (long long)B * nblk * kLossSums;
    hipStream_t st = (hipStream_t)stream;
    // the 12 loss parameters travel through a small device buffer at the tail of `losses` (4 + 12 floats)
    SK_CHECK_HIP(hipMemcpyAsync(losses + 4, loss_params_host, 12 * sizeof(float), hipMemcpyHostToDevice, st));
    loss_reduce_kernel<<<dim3(nblk, B), 256, 0, st>>>(a, nblk, partial);
    SK_CHECK_LAUNCH();
    loss_finalize_kernel<<<1, 256, 0, st>>>(partial, B, nblk, losses + 4, losses, coef);
    SK_CHECK_LAUNCH();
    if (dlogits) {
        loss_bwd_kernel<<<dim3(sk::stream_grid(a.n, 256, 2), B), 256, 0, st>>>(a, coef, dlogits);
        SK_CHECK_LAUNCH();
    }
    return SK_OK;
}

int sk_baked_embed_to_prob(const float* embedding, const float* baked, float* out, int B, int64_t voxels,
                           const float* sigma_host, float eps, void* stream) {
    SK_CHECK_ARG(embedding && baked && out && sigma_host && B >= 1 && voxels >= 1, "sk_baked_embed_to_prob: bad arguments");
    float iv[3];
    for (int k = 0; k < 3; ++k) {
        const float sg = sigma_host[k] + eps;
        iv[k] = 1.0f / (sg * sg * 2.0f * -1.0f);
    }
    embed_prob_kernel<<<dim3(sk::stream_grid(voxels, 256, 2), B), 256, 0, (hipStream_t)stream>>>(embedding, baked, out, voxels,
                                                                                                  iv[0], iv[1], iv[2]);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_train_tversky(const float* predicted, const float* ground_truth, int B, int64_t voxels, float alpha, float beta,
                     float eps, float* loss, float* workspace, void* stream) {
    SK_CHECK_ARG(predicted && ground_truth && loss && workspace && B >= 1 && voxels >= 1, "sk_train_tversky: bad arguments");
    const int nblk = sk_train_loss_num_blocks(voxels);
    tversky_reduce_kernel<<<dim3(nblk, B), 256, 0, (hipStream_t)stream>>>(predicted, ground_truth, voxels, nblk, workspace);
    SK_CHECK_LAUNCH();
    tversky_finalize_kernel<<<1, 256, 0, (hipStream_t)stream>>>(workspace, B, nblk, alpha, beta, eps, loss);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

static int wgrad_plan(int B, int ox, int oy, int oz, int cout, int cin, int ksize, int* nchunk, int* nchunk_b,
                      long long* chunk, int* ngroup) {
    const int ncot = (cout + 31) / 32, ncit = (cin + 31) / 32;
    *ngroup = ksize == 3 ? 3 : 1;
    const long long nvox = (long long)ox * oy * oz;
    long long want = 4096 / ((long long)ncot * ncit * *ngroup * B);  // chunks per batch item: ~4096 waves in all
    if (want < 1) want = 1;
    long long c = (nvox + want - 1) / want;
    if (c < 512) c = 512;
    c = (c + 1) & ~1LL;
    *chunk = c;
    *nchunk_b = (int)((nvox + c - 1) / c);
    *nchunk = B * *nchunk_b;
    return 0;
}

// wgrad16x_kernel: 16 x 16 (y, z) footprints x segments of x, about 1024 workgroups (four rounds of the 256 CUs) but
// at least eight planes a segment (each segment fetches two planes it does not own).  Returns the chunk count, 0 if the
// kernel does not cover the shape.
static int wgrad_x_plan(int B, int ox, int oy, int oz, int cout, int cin, int ksize, int* nxs, int* xseg) {
    if (ksize != 3 || oy % 16 || oz % 16 || cout % 32 || cin % 32) return 0;
    const long long wg = (long long)B * (oy / 16) * (oz / 16) * (cout / 32) * (cin / 32);
    long long s = (1024 + wg - 1) / wg;
    if (s > ox / 8) s = ox / 8;
    if (s < 1) s = 1;
    *xseg = (int)((ox + s - 1) / s);
    *nxs = (ox + *xseg - 1) / *xseg;
    return B * (oy / 16) * (oz / 16) * *nxs;
}

int64_t sk_train_conv_wgrad_workspace_floats(int B, int ox, int oy, int oz, int cout, int cin, int ksize) {
    int nchunk, nchunk_b, ngroup, nxs, xseg;
    long long chunk;
    wgrad_plan(B, ox, oy, oz, cout, cin, ksize, &nchunk, &nchunk_b, &chunk, &ngroup);
    const int nx = wgrad_x_plan(B, ox, oy, oz, cout, cin, ksize, &nxs, &xseg);
    if (nx > nchunk) nchunk = nx;
    return (int64_t)nchunk * ((int64_t)cout * cin * ksize * ksize * ksize + cout);
}

int sk_train_conv_wgrad(const sk_conv_src* srcs, int n_src, const float* dy, int B, int ox, int oy, int oz, int cout,
                        int ksize, float* dweight, float* dbias, float* workspace, void* stream) {
    SK_CHECK_ARG(srcs && dy && dweight && workspace, "sk_train_conv_wgrad: NULL pointer");
    SK_CHECK_ARG(n_src == 1 || n_src == 2, "sk_train_conv_wgrad: n_src must be 1 or 2");
    SK_CHECK_ARG(ksize == 1 || ksize == 2 || ksize == 3, "sk_train_conv_wgrad: ksize must be 1, 2 or 3");
    SK_CHECK_ARG(B >= 1 && ox >= 1 && oy >= 1 && oz >= 1 && cout >= 1, "sk_train_conv_wgrad: bad extents");
    WgradArgs a{};
    a.nsrc = n_src;
    for (int i = 0; i < n_src; ++i) {
        SK_CHECK_ARG(srcs[i].data && srcs[i].c > 0 && srcs[i].affine == nullptr, "sk_train_conv_wgrad: bad source %d", i);
        SK_CHECK_ARG(n_src == 1 || srcs[i].c % 32 == 0, "sk_train_conv_wgrad: two sources need c %% 32 == 0");
        int up = srcs[i].upsample ? 1 : 0;
        SK_CHECK_ARG(!up || (ksize == 3 && ox % 2 == 0 && oy % 2 == 0 && oz % 2 == 0),
                     "sk_train_conv_wgrad: upsampled source needs ksize 3 and even output extents");
        a.src[i].data = (const float*)srcs[i].data;
        a.src[i].C = srcs[i].c;
        a.src[i].up = up;
        int s = (ksize == 3) ? 1 : ksize;
        a.src[i].Xs = up ? ox / 2 : ox * s;
        a.src[i].Ys = up ? oy / 2 : oy * s;
        a.src[i].Zs = up ? oz / 2 : oz * s;
        a.cin += srcs[i].c;
    }
    a.dy = dy;
    a.B = B;
    a.ox = ox;
    a.oy = oy;
    a.oz = oz;
    a.cout = cout;
    a.ksize = ksize;
    a.ncot = (cout + 31) / 32;
    a.ncit = (a.cin + 31) / 32;
    wgrad_plan(B, ox, oy, oz, cout, a.cin, ksize, &a.nchunk, &a.nchunk_b, &a.chunk, &a.ngroup);
    SK_CHECK_ARG((long long)ox * oy * oz * cout * 4 < (1LL << 32), "sk_train_conv_wgrad: dy of one batch item must be < 4 GiB");
    for (int i = 0; i < n_src; ++i)
        SK_CHECK_ARG((long long)a.src[i].Xs * a.src[i].Ys * a.src[i].Zs * a.src[i].C * 4 < (1LL << 32),
                     "sk_train_conv_wgrad: source %d of one batch item must be < 4 GiB", i);
    const long long nw = (long long)cout * a.cin * ksize * ksize * ksize;
    a.part = workspace;
    a.part_bias = dbias ? workspace + (long long)a.nchunk * nw : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((long long)a.nchunk * a.ncot * a.ncit * a.ngroup);
    const bool stem = ksize == 3 && n_src == 1 && a.cin == 1 && !a.src[0].up;
    if (stem)
        wgrad_stem_kernel<false><<<(unsigned)((long long)a.nchunk * a.ncot), 64, 0, st>>>(a);
    else if (ksize == 3)
        wgrad_kernel<9><<<grid, 64, 0, st>>>(a);
    else if (ksize == 2)
        wgrad_kernel<8><<<grid, 64, 0, st>>>(a);
    else
        wgrad_kernel<1><<<grid, 64, 0, st>>>(a);
    SK_CHECK_LAUNCH();
    wgrad_reduce_kernel<<<sk::cdiv(nw, 64), 64 * kWredSlices, 0, st>>>(a.part, a.nchunk, nw, dweight, nullptr, cout, a.cin, stem ? 0 : ksize * ksize * ksize);
    SK_CHECK_LAUNCH();
    if (dbias) {
        wgrad_reduce_kernel<<<sk::cdiv(cout, 64), 64 * kWredSlices, 0, st>>>(a.part_bias, a.nchunk, cout, dbias, nullptr, 0, 0, 0);
        SK_CHECK_LAUNCH();
    }
    return SK_OK;
}

int sk_train_stem_wgrad_f16(const float* image, const void* dy16, const float* dy_scale, int B, int X, int Y, int Z,
                            float* dweight, float* dbias, float* workspace, void* stream) {
    SK_CHECK_ARG(image && dy16 && dy_scale && dweight && workspace, "sk_train_stem_wgrad_f16: NULL pointer");
    SK_CHECK_ARG(B >= 1 && X >= 1 && Y >= 1 && Z >= 1, "sk_train_stem_wgrad_f16: bad extents");
    const int cout = 32;
    WgradArgs a{};
    a.nsrc = 1;
    a.src[0].data = image;
    a.src[0].C = 1;
    a.src[0].up = 0;
    a.src[0].Xs = X;
    a.src[0].Ys = Y;
    a.src[0].Zs = Z;
    a.dy = reinterpret_cast<const float*>(dy16);
    a.B = B;
    a.ox = X;
    a.oy = Y;
    a.oz = Z;
    a.cout = cout;
    a.cin = 1;
    a.ksize = 3;
    a.ncot = 1;
    a.ncit = 1;
    wgrad_plan(B, X, Y, Z, cout, 1, 3, &a.nchunk, &a.nchunk_b, &a.chunk, &a.ngroup);
    SK_CHECK_ARG((long long)X * Y * Z * cout * 2 < (1LL << 32), "sk_train_stem_wgrad_f16: dy of one batch item must be < 4 GiB");
    const long long nw = (long long)cout * 27;
    a.part = workspace;
    a.part_bias = dbias ? workspace + (long long)a.nchunk * nw : nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (Z % 16 == 0) {
        a.chunk = (a.chunk + 15) & ~15LL;   // whole 16-voxel steps; the plan's chunk count still covers the volume
        wgrad_stem16_kernel<<<(unsigned)a.nchunk, 64, 0, st>>>(a);
    } else
        wgrad_stem_kernel<true><<<(unsigned)((long long)a.nchunk * a.ncot), 64, 0, st>>>(a);
    SK_CHECK_LAUNCH();
    wgrad_reduce_kernel<<<sk::cdiv(nw, 64), 64 * kWredSlices, 0, st>>>(a.part, a.nchunk, nw, dweight, dy_scale, cout, 1, 0);
    SK_CHECK_LAUNCH();
    if (dbias) {
        wgrad_reduce_kernel<<<sk::cdiv(cout, 64), 64 * kWredSlices, 0, st>>>(a.part_bias, a.nchunk, cout, dbias, dy_scale, 0, 0, 0);
        SK_CHECK_LAUNCH();
    }
    return SK_OK;
}

int sk_train_conv_wgrad_f16(const sk_conv_src* srcs, int n_src, const void* dy, const float* dy_scale, int B, int ox, int oy,
                            int oz, int cout, int ksize, float* dweight, float* dbias, float* workspace,
                            const void* zero_page, void* stream) {
    SK_CHECK_ARG(srcs && dy && dweight && workspace, "sk_train_conv_wgrad_f16: NULL pointer");
    SK_CHECK_ARG(n_src == 1 || n_src == 2, "sk_train_conv_wgrad_f16: n_src must be 1 or 2");
    SK_CHECK_ARG(ksize == 1 || ksize == 2 || ksize == 3, "sk_train_conv_wgrad_f16: ksize must be 1, 2 or 3");
    SK_CHECK_ARG(B >= 1 && ox >= 1 && oy >= 1 && oz >= 1 && cout >= 1, "sk_train_conv_wgrad_f16: bad extents");
    Wgrad16Args a{};
    a.nsrc = n_src;
    for (int i = 0; i < n_src; ++i) {
        SK_CHECK_ARG(srcs[i].data && srcs[i].c > 0 && srcs[i].affine == nullptr, "sk_train_conv_wgrad_f16: bad source %d", i);
        SK_CHECK_ARG(n_src == 1 || srcs[i].c % 32 == 0, "sk_train_conv_wgrad_f16: two sources need c %% 32 == 0");
        int up = srcs[i].upsample ? 1 : 0;
        SK_CHECK_ARG(!up || (ksize == 3 && ox % 2 == 0 && oy % 2 == 0 && oz % 2 == 0),
                     "sk_train_conv_wgrad_f16: upsampled source needs ksize 3 and even output extents");
        a.src[i].data = (const t16*)srcs[i].data;
        a.src[i].C = srcs[i].c;
        a.src[i].up = up;
        int s = (ksize == 3) ? 1 : ksize;
        a.src[i].Xs = up ? ox / 2 : ox * s;
        a.src[i].Ys = up ? oy / 2 : oy * s;
        a.src[i].Zs = up ? oz / 2 : oz * s;
        a.cin += srcs[i].c;
        SK_CHECK_ARG((long long)a.src[i].Xs * a.src[i].Ys * a.src[i].Zs * a.src[i].C * 2 < (1LL << 32),
                     "sk_train_conv_wgrad_f16: source %d of one batch item must be < 4 GiB", i);
    }
    SK_CHECK_ARG((long long)ox * oy * oz * cout * 2 < (1LL << 32), "sk_train_conv_wgrad_f16: dy of one batch item must be < 4 GiB");
    a.dy = (const t16*)dy;
    a.B = B;
    a.ox = ox;
    a.oy = oy;
    a.oz = oz;
    a.cout = cout;
    a.ksize = ksize;
    a.ncot = (cout + 31) / 32;
    a.ncit = (a.cin + 31) / 32;
    wgrad_plan(B, ox, oy, oz, cout, a.cin, ksize, &a.nchunk, &a.nchunk_b, &a.chunk, &a.ngroup);
    a.chunk = (a.chunk + 15) & ~15LL;  // whole 16-voxel MFMA steps; the chunk count of the plan still covers the volume
    const long long nw = (long long)cout * a.cin * ksize * ksize * ksize;
    a.part = workspace;
    a.part_bias = dbias ? workspace + (long long)a.nchunk * nw : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((long long)a.nchunk * a.ncot * a.ncit * a.ngroup);
    bool lines = zero_page != nullptr && cout % 32 == 0;  // whole 64-byte channel lines: LDS-DMA + transposed reads
    for (int i = 0; i < n_src; ++i) lines = lines && srcs[i].c % 32 == 0;
    bool strips = true;   // -DSK_TUNING builds (tools/) can switch the strip kernel off for A/B timing
#ifdef SK_TUNING
    strips = getenv("SK_WGRAD_NOSTRIP") == nullptr;
#endif
    int nxs = 0, xseg = 0;
    int nchunk_x = lines && strips ? wgrad_x_plan(B, ox, oy, oz, cout, a.cin, ksize, &nxs, &xseg) : 0;
#ifdef SK_TUNING
    if (getenv("SK_WGRAD_NOXMARCH")) nchunk_x = 0;
#endif
    if (nchunk_x > 0) {
#ifdef SK_TIMING
        a.dbg = sk::timing_buffer();
#endif
        a.nchunk = nchunk_x;
        a.part_bias = dbias ? workspace + (long long)a.nchunk * nw : nullptr;
        SK_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad16x_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kWxLds));
        const unsigned gx = (unsigned)((((long long)a.nchunk + 7) / 8) * 8 * a.ncot * a.ncit);   // whole XCD rounds of chunks
        wgrad16x_kernel<<<gx, 256, kWxLds, st>>>(a, oy / 16, oz / 16, nxs, xseg);
    } else if (lines && ksize == 3 && oz % 16 == 0 && oy % 4 == 0 && strips) {
        a.ngroup = 3;
        const unsigned g3 = (unsigned)((((long long)a.nchunk + 7) / 8) * 8 * a.ncot * a.ncit * 3);   // whole XCD rounds of chunks
        const int ncol = ox * (oz / 16);
        wgrad16y_kernel<<<g3, 64, 0, st>>>(a, (const char*)zero_page, (ncol + a.nchunk_b - 1) / a.nchunk_b);
    } else if (lines) {
        if (ksize == 3) {
            // three taps per wave (nine tap groups): 8 KiB of LDS tiles per wave instead of 20, so ~2.5x the waves
            // and DMA steps in flight per CU -- the kernel is bound by the latency of its tile loads
            int tpw = 9;  // 3 measured 1.5 % slower: the kernel is bound by L2 bandwidth (every input line is read once per tap), not by latency
#ifdef SK_TUNING
            if (const char* e = getenv("SK_WGRAD_TPW")) tpw = atoi(e) == 3 ? 3 : 9;
#endif
            a.ngroup = 27 / tpw;
            const unsigned g = (unsigned)((long long)a.nchunk * a.ncot * a.ncit * a.ngroup);
            if (tpw == 3)
                wgrad16t_kernel<3><<<g, 64, 0, st>>>(a, (const char*)zero_page);
            else
                wgrad16t_kernel<9><<<g, 64, 0, st>>>(a, (const char*)zero_page);
        }
        else if (ksize == 2)
            wgrad16t_kernel<8><<<grid, 64, 0, st>>>(a, (const char*)zero_page);
        else
            wgrad16t_kernel<1><<<grid, 64, 0, st>>>(a, (const char*)zero_page);
    } else if (ksize == 3)
        wgrad16_kernel<9><<<grid, 64, 0, st>>>(a);
    else if (ksize == 2)
        wgrad16_kernel<8><<<grid, 64, 0, st>>>(a);
    else
        wgrad16_kernel<1><<<grid, 64, 0, st>>>(a);
    SK_CHECK_LAUNCH();
    wgrad_reduce_kernel<<<sk::cdiv(nw, 64), 64 * kWredSlices, 0, st>>>(a.part, a.nchunk, nw, dweight, dy_scale, cout, a.cin, ksize * ksize * ksize);
    SK_CHECK_LAUNCH();
    if (dbias) {
        wgrad_reduce_kernel<<<sk::cdiv(cout, 64), 64 * kWredSlices, 0, st>>>(a.part_bias, a.nchunk, cout, dbias, dy_scale, 0, 0, 0);
        SK_CHECK_LAUNCH();
    }
    return SK_OK;
}

int sk_train_pack_weight(const float* weight, int Co, int Ci, int ksize, int transposed, int c_lo, int c_n, void* dst,
                         void* stream) {
    SK_CHECK_ARG(weight && dst && (ksize == 1 || ksize == 2 || ksize == 3), "sk_train_pack_weight: bad arguments");
    SK_CHECK_ARG(transposed >= 0 && transposed <= 2, "sk_train_pack_weight: transposed must be 0, 1 or 2");
    const int eff_cout = transposed ? c_n : Co, eff_cin = transposed ? Co : Ci;
    SK_CHECK_ARG(!transposed || (c_lo >= 0 && c_n >= 1 && c_lo + c_n <= Ci), "sk_train_pack_weight: bad channel range");
    SK_CHECK_ARG(transposed || (c_lo == 0 && c_n == Ci), "sk_train_pack_weight: a channel range needs transposed");
    SK_CHECK_ARG(eff_cout % 32 == 0 && eff_cin % (ksize == 3 ? 32 : 16) == 0,
                 "sk_train_pack_weight: unsupported shape cout=%d cin=%d k=%d", eff_cout, eff_cin, ksize);
    PackArgs a{};
    a.w = weight;
    a.dst = (t16*)dst;
    a.Co = Co;
    a.Ci = Ci;
    a.ksize = ksize;
    a.eff_cout = eff_cout;
    a.eff_cin = eff_cin;
    a.transposed = transposed;
    a.c_lo = c_lo;
    a.n = (long long)ksize * ksize * ksize * (eff_cin / 16) * (eff_cout / 32) * 512;
    pack_weight_kernel<<<sk::cdiv(a.n, 256), 256, 0, (hipStream_t)stream>>>(a);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

static int interleave2_impl(const void* tens16, const void* add16, const float* add_scale, float* dx, int B, int cx, int cy, int cz,
                            int C, const float* scale, int accumulate, void* stream) {
    SK_CHECK_ARG(tens16 && dx && B >= 1 && cx >= 1 && cy >= 1 && cz >= 1 && C >= 8 && C % 8 == 0,
                 "sk_train_interleave2: bad arguments (C must be a multiple of 8)");
    SK_CHECK_ARG(!add16 || add_scale, "sk_train_interleave2_add16: add_scale is NULL");
    long long n = (long long)B * cx * cy * cz * C;   // fine voxels x channel octets = coarse voxels x C
    interleave2_kernel<<<sk::stream_grid(n, 256, 2), 256, 0, (hipStream_t)stream>>>((const t16*)tens16, dx, B, cx, cy, cz, C, scale,
                                                                                    accumulate ? 1 : 0, (const t16*)add16,
                                                                                    add_scale);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_train_interleave2(const void* tens16, float* dx, int B, int cx, int cy, int cz, int C, const float* scale, int accumulate,
                         void* stream) {
    return interleave2_impl(tens16, nullptr, nullptr, dx, B, cx, cy, cz, C, scale, accumulate, stream);
}

int sk_train_interleave2_add16(const void* tens16, const void* add16, const float* add_scale, float* dx, int B, int cx, int cy,
                               int cz, int C, const float* scale, void* stream) {
    SK_CHECK_ARG(add16, "sk_train_interleave2_add16: add16 is NULL");
    return interleave2_impl(tens16, add16, add_scale, dx, B, cx, cy, cz, C, scale, 0, stream);
}

int sk_train_interleave2_h(const void* tens16, const void* add16, const float* add_scale, void* dx16, float* out_scale, int B,
                           int cx, int cy, int cz, int C, const float* scale, void* stream) {
    SK_CHECK_ARG(tens16 && dx16 && out_scale && scale && (!add16 || add_scale), "sk_train_interleave2_h: NULL pointer");
    SK_CHECK_ARG(B >= 1 && cx >= 1 && cy >= 1 && cz >= 1 && C >= 8 && C % 8 == 0, "sk_train_interleave2_h: bad extents (C %% 8 == 0)");
    const long long n = (long long)B * cx * cy * cz * 8 * (C / 8);
    interleave2_h_kernel<<<sk::stream_grid(n, 256, 4), 256, 0, (hipStream_t)stream>>>((const t16*)tens16, (t16*)dx16, B, cx, cy, cz, C,
                                                                                    scale, (const t16*)add16, add_scale, out_scale);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_train_sumpool2_hh(const void* fine16, const float* scale, void* coarse16, float* out_scale, int B, int cx, int cy, int cz,
                         int C, void* stream) {
    SK_CHECK_ARG(fine16 && scale && coarse16 && out_scale, "sk_train_sumpool2_hh: NULL pointer");
    SK_CHECK_ARG(B >= 1 && cx >= 1 && cy >= 1 && cz >= 1 && C >= 8 && C % 8 == 0, "sk_train_sumpool2_hh: bad extents (C %% 8 == 0)");
    const long long n = (long long)B * cx * cy * cz * (C / 8);
    sumpool2_hh_kernel<<<sk::stream_grid(n, 256, 2), 256, 0, (hipStream_t)stream>>>((const t16*)fine16, scale, (t16*)coarse16,
                                                                                  out_scale, B, cx, cy, cz, C);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_train_heads_dgrad_f16(const float* dlogits, const float* dl_scale, const float* weight, void* dx16, float* out_scale,
                             int64_t nvox, int C, void* stream) {
    SK_CHECK_ARG(dlogits && dl_scale && weight && dx16 && out_scale && nvox >= 1, "sk_train_heads_dgrad_f16: bad arguments");
    SK_CHECK_ARG(C >= 8 && C % 8 == 0 && 256 % (C / 8) == 0 && C <= 256, "sk_train_heads_dgrad_f16: C=%d unsupported", C);
    heads_dgrad_h_kernel<5><<<sk::stream_grid(nvox * (C / 8), 256, 4), 256, 0, (hipStream_t)stream>>>(
        dlogits, weight, (t16*)dx16, out_scale, dl_scale, (long long)nvox, C);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_train_gn_silu_f16(const void* y16, const float* affine, void* z16, float* z32, int B, int64_t voxels, int C,
                         void* stream) {
    SK_CHECK_ARG(y16 && affine && z16 && C > 0, "sk_train_gn_silu_f16: bad arguments");
    long long n = voxels * C;
    SK_CHECK_ARG(C % 8 == 0 && 256 % (C / 8) == 0, "sk_train_gn_silu_f16: C=%d unsupported", C);
    gn_silu_f16_kernel<<<dim3(sk::stream_grid(n / 8, 256, 4), B), 256, 0, (hipStream_t)stream>>>((const t16*)y16, affine,
                                                                                             (t16*)z16, z32, C, n);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int64_t sk_train_gn_bwd_f16_workspace_floats(int B, int64_t voxels, int C) {
    return (int64_t)B * sk_train_gn_bwd_num_blocks(voxels) * C * 4 + (int64_t)B * C * 3;
}

static int gn_silu_bwd_f16_impl(const void* dz, const float* dz_scale, const void* y16, const float* affine,
                                const float* stats, const float* gamma, int B, int64_t voxels, int C, int groups,
                                void* dy16, float* scale, float* dgamma, float* dbeta, float* workspace, void* stream) {
    SK_CHECK_ARG(dz && y16 && affine && stats && gamma && dy16 && scale && dgamma && dbeta && workspace,
                 "sk_train_gn_silu_bwd_f16: NULL pointer");
    SK_CHECK_ARG((C == 32 || C == 64 || C == 128) && groups > 0 && groups <= 16 && C % groups == 0,
                 "sk_train_gn_silu_bwd_f16: C=%d groups=%d unsupported", C, groups);
    const int nblk = gnb_blocks(voxels, C);                  // <= sk_train_gn_bwd_num_blocks: the workspace fits
    float* partial = workspace;                              // (B, nblk, C, 4)
    float* coef = workspace + (long long)B * nblk * C * 4;   // (B, C, 3)
    hipStream_t st = (hipStream_t)stream;
    if (dz_scale)
        gn_bwd_reduce16_kernel<true><<<dim3(nblk, B), 256, 0, st>>>(dz, dz_scale, (const t16*)y16, affine, stats, C, groups,
                                                                    voxels, nblk, partial);
    else
        gn_bwd_reduce16_kernel<false><<<dim3(nblk, B), 256, 0, st>>>(dz, nullptr, (const t16*)y16, affine, stats, C, groups,
                                                                     voxels, nblk, partial);
    SK_CHECK_LAUNCH();
    gn_bwd_finalize16_kernel<<<1, 1024, 0, st>>>(partial, B, nblk, C, groups, (double)voxels, gamma, stats, coef, dgamma, dbeta,
                                                 scale);
    SK_CHECK_LAUNCH();
    long long n = voxels * C;
    const dim3 grid(sk::stream_grid(n / 8, 256, 4), B);
    if (dz_scale)
        gn_bwd_apply16_kernel<true><<<grid, 256, 0, st>>>(dz, dz_scale, (const t16*)y16, affine, stats, coef, scale,
                                                          (t16*)dy16, C, groups, n);
    else
        gn_bwd_apply16_kernel<false><<<grid, 256, 0, st>>>(dz, nullptr, (const t16*)y16, affine, stats, coef, scale,
                                                           (t16*)dy16, C, groups, n);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_train_gn_silu_bwd_f16(const float* dz, const void* y16, const float* affine, const float* stats, const float* gamma,
                             int B, int64_t voxels, int C, int groups, void* dy16, float* scale, float* dgamma, float* dbeta,
                             float* workspace, void* stream) {
    return gn_silu_bwd_f16_impl(dz, nullptr, y16, affine, stats, gamma, B, voxels, C, groups, dy16, scale, dgamma, dbeta,
                                workspace, stream);
}

int sk_train_gn_silu_bwd_f16h(const void* dz16, const float* dz_scale, const void* y16, const float* affine,
                              const float* stats, const float* gamma, int B, int64_t voxels, int C, int groups, void* dy16,
                              float* scale, float* dgamma, float* dbeta, float* workspace, void* stream) {
    SK_CHECK_ARG(dz_scale, "sk_train_gn_silu_bwd_f16h: dz_scale is NULL");
    return gn_silu_bwd_f16_impl(dz16, dz_scale, y16, affine, stats, gamma, B, voxels, C, groups, dy16, scale, dgamma, dbeta,
                                workspace, stream);
}

int sk_train_absmax_scale(const float* x, int64_t n, float* scale, void* stream) {
    SK_CHECK_ARG(x && scale && n >= 1, "sk_train_absmax_scale: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    unsigned* mx = reinterpret_cast<unsigned*>(scale + 2);  // scale: 3 floats; [2] is the max's bit pattern
    SK_CHECK_HIP(hipMemsetAsync(mx, 0, sizeof(unsigned), st));
    absmax_kernel<<<sk::stream_grid(n, 256, 8), 256, 0, st>>>(x, n, mx);
    SK_CHECK_LAUNCH();
    scale_from_absmax_kernel<<<1, 1, 0, st>>>(mx, scale);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_train_cast_f32_f16(const float* x, void* y, int64_t n, const float* scale, void* stream) {
    SK_CHECK_ARG(x && y && n >= 1, "sk_train_cast_f32_f16: bad arguments");
    const int vec = n % 8 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
    cast_f32_f16_kernel<<<sk::stream_grid(vec ? n / 8 : n, 256, 4), 256, 0, (hipStream_t)stream>>>(x, (t16*)y, n, scale, vec);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_train_cast_f16_f32(const void* x, float* y, int64_t n, const float* scale, int accumulate, void* stream) {
    SK_CHECK_ARG(x && y && n >= 1, "sk_train_cast_f16_f32: bad arguments");
    const int vec = n % 8 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
    cast_f16_f32_kernel<<<sk::stream_grid(vec ? n / 8 : n, 256, 4), 256, 0, (hipStream_t)stream>>>((const t16*)x, y, n, scale,
                                                                                     accumulate ? 1 : 0, vec);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_train_sumpool2(const float* fine, float* coarse, int B, int cx, int cy, int cz, int C, void* stream) {
    SK_CHECK_ARG(fine && coarse && B >= 1 && cx >= 1 && cy >= 1 && cz >= 1 && C >= 1, "sk_train_sumpool2: bad arguments");
    long long n = (long long)B * cx * cy * cz * C;
    sumpool2_kernel<<<sk::stream_grid(n, 256, 2), 256, 0, (hipStream_t)stream>>>(fine, coarse, B, cx, cy, cz, C);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_train_sumpool2_f16(const void* fine16, const float* scale, float* coarse, int B, int cx, int cy, int cz, int C,
                          void* stream) {
    SK_CHECK_ARG(fine16 && scale && coarse && B >= 1 && cx >= 1 && cy >= 1 && cz >= 1 && C >= 8 && C % 8 == 0,
                 "sk_train_sumpool2_f16: bad arguments");
    long long n = (long long)B * cx * cy * cz * (C / 8);
    sumpool2_f16_kernel<<<sk::stream_grid(n, 256, 2), 256, 0, (hipStream_t)stream>>>((const t16*)fine16, scale, coarse, B, cx, cy,
                                                                                     cz, C);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_train_heads_fwd_f16(const void* z16, const float* weight, const float* bias, float* logits, int64_t nvox,
                           void* stream) {
    SK_CHECK_ARG(z16 && weight && bias && logits && nvox >= 1, "sk_train_heads_fwd_f16: bad arguments");
    heads_fwd_f16_kernel<<<sk::stream_grid(nvox, 64, 4), 256, 0, (hipStream_t)stream>>>((const t16*)z16, weight, bias, logits,
                                                                                      nvox);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int64_t sk_train_heads_wgrad_workspace_floats(int64_t nvox) { return ((nvox + kHeadsVox - 1) / kHeadsVox) * 165; }

int sk_train_heads_wgrad_f16(const void* z16, const float* dlogits, float* dweight, float* dbias, int64_t nvox,
                             float* workspace, void* stream) {
    SK_CHECK_ARG(z16 && dlogits && dweight && dbias && workspace && nvox >= 1, "sk_train_heads_wgrad_f16: bad arguments");
    const int nb = (int)((nvox + kHeadsVox - 1) / kHeadsVox);
    float* part = workspace;
    float* part_b = workspace + (long long)nb * 160;
    hipStream_t st = (hipStream_t)stream;
    heads_wgrad_f16_kernel<<<nb, 256, 0, st>>>((const t16*)z16, dlogits, part, part_b, nvox);
    SK_CHECK_LAUNCH();
    wgrad_reduce_kernel<<<sk::cdiv(160, 64), 64 * kWredSlices, 0, st>>>(part, nb, 160, dweight, nullptr, 0, 0, 0);
    SK_CHECK_LAUNCH();
    wgrad_reduce_kernel<<<1, 64 * kWredSlices, 0, st>>>(part_b, nb, 5, dbias, nullptr, 0, 0, 0);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_train_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, void* stream) {
    SK_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "sk_train_adamw: bad arguments");
    if (n == 0) return SK_OK;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    adamw_kernel<<<sk::stream_grid(n, 256, 2), 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                                                                             beta2, eps, weight_decay, (float)bc1,
                                                                             (float)sqrt(bc2));
    SK_CHECK_LAUNCH();
    return SK_OK;
}

}  // extern "C"

