// Stage 3 of skoots.lib.eval.eval(): offset following + skeleton-label assignment.
//
// Replaces (reference file:line)
//   skoots/lib/vector_to_embedding.py:79-132   _vec2embed3D
//   skoots/lib/skeleton.py:656-695             index_skeleton_by_embed
//   skoots/lib/eval.py:245-284                 the per-crop loop around them
//
// One thread per voxel; the N-1 dependent lookups are a pointer chase kept in
// registers.  Arithmetic is the reference's, operation for operation, in fp32 with
// no FMA contraction (this file is compiled with -ffp-contract=off and uses the
// explicit _rn intrinsics): half-to-even rounding, per-axis clamp to [0, k]
// INCLUSIVE of k, fp32 ravel, clamp of the flat index, gather by flat index (so an
// index clamped to k wraps into the next row / plane exactly as `take` does).
//
// HBM-bound / latency-bound integer-ish work: no MFMA here.  The fused kernel reads
// vectors from an interleaved (X,Y,Z,4) fp16 volume so that one hop is one 8-byte
// load; the crop overcompute of the reference (2.7-6x) disappears because each
// voxel is evaluated once, inside the window of the crop that writes it last.
#include "common.h"

namespace {

constexpr int kMaxIter = 64;

struct FollowParams {
    int w, h, d;        // crop-local extents (effective crop)
    int n_iter;         // N (>= 1); N-1 lookups
    float fw, fh, fd;   // float(w), float(h), float(d)
    float fmax_flat;    // float(w*h*d - 1)
    long long nvox;     // w*h*d
    float sc[kMaxIter][3];  // sc[0] = float(scale); sc[i] = float(decay^i) * float(scale)
};

__device__ __forceinline__ float clampf(float v, float lo, float hi) {
    // torch.clamp == min(max(v, lo), hi); NaN handling is irrelevant (tanh outputs).
    return fminf(fmaxf(v, lo), hi);
}

__device__ __forceinline__ long long flat_index(const FollowParams& p, float mx, float my,
                                                float mz, int& qx, int& qy, int& qz) {
    float fx = clampf(rintf(mx), 0.0f, p.fw);  // vector_to_embedding.py:117-119
    float fy = clampf(rintf(my), 0.0f, p.fh);
    float fz = clampf(rintf(mz), 0.0f, p.fd);
    qx = (int)fx;
    qy = (int)fy;
    qz = (int)fz;
    // :122-126 (index0 * y * z) + (index1 * z) + index2, every product/sum rounded to fp32
    float t = __fmul_rn(__fmul_rn(fx, p.fh), p.fd);
    float u = __fmul_rn(fy, p.fd);
    float f = __fadd_rn(__fadd_rn(t, u), fz);
    f = clampf(f, 0.0f, p.fmax_flat);  // :127
    long long idx = (long long)f;
    if (idx > p.nvox - 1) idx = p.nvox - 1;  // only reachable when fmax_flat rounded up
    if (idx < 0) idx = 0;
    return idx;
}

// ---------------------------------------------------------------- library kernel (one crop)
template <typename T>
__device__ __forceinline__ float ldv(const T* p, long long i);
template <>
__device__ __forceinline__ float ldv<__half>(const __half* p, long long i) {
    return __half2float(p[i]);
}
template <>
__device__ __forceinline__ float ldv<float>(const float* p, long long i) {
    return p[i];
}

template <typename T>
__global__ void __launch_bounds__(256) vec2embed_kernel(const T* __restrict__ vec,
                                                         float* __restrict__ embed,
                                                         FollowParams p) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.nvox) return;
    int hd = p.h * p.d;
    int x = (int)(i / hd);
    int r = (int)(i - (long long)x * hd);
    int y = r / p.d;
    int z = r - y * p.d;
    float v0 = ldv(vec, i), v1 = ldv(vec, i + p.nvox), v2 = ldv(vec, i + 2 * p.nvox);
    float mx = __fadd_rn((float)x, __fmul_rn(v0, p.sc[0][0]));  // :104-105
    float my = __fadd_rn((float)y, __fmul_rn(v1, p.sc[0][1]));
    float mz = __fadd_rn((float)z, __fmul_rn(v2, p.sc[0][2]));
    for (int it = 1; it < p.n_iter; ++it) {
        int qx, qy, qz;
        long long j = flat_index(p, mx, my, mz, qx, qy, qz);
        float g0 = ldv(vec, j), g1 = ldv(vec, j + p.nvox), g2 = ldv(vec, j + 2 * p.nvox);
        mx = __fadd_rn(mx, __fmul_rn(g0, p.sc[it][0]));  // :129-130
        my = __fadd_rn(my, __fmul_rn(g1, p.sc[it][1]));
        mz = __fadd_rn(mz, __fmul_rn(g2, p.sc[it][2]));
    }
    embed[i] = mx;
    embed[i + p.nvox] = my;
    embed[i + 2 * p.nvox] = mz;
}

// ---------------------------------------------------------------- gather kernel
template <typename L>
__global__ void __launch_bounds__(256) index_by_embed_kernel(const L* __restrict__ labels, int lx,
                                                              int ly, int lz,
                                                              const float* __restrict__ embed,
                                                              long long n,
                                                              int32_t* __restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // skeleton.py:678-683
    int xi = (int)clampf(rintf(embed[i]), 0.0f, (float)(lx - 1));
    int yi = (int)clampf(rintf(embed[i + n]), 0.0f, (float)(ly - 1));
    int zi = (int)clampf(rintf(embed[i + 2 * n]), 0.0f, (float)(lz - 1));
    out[i] = (int32_t)labels[((long long)xi * ly + yi) * lz + zi];
}

// ---------------------------------------------------------------- fused stage-3 kernel
struct AssignGeom {
    int X, Y, Z;          // global volume (the label volume)
    int win_lo, win_hi;   // z-window held by the vec4 / out arrays (Z-sharding: slab + halo)
    int z_lo, z_hi;       // planes written by this launch (global z)
};

__device__ __forceinline__ void unpack_vec4(uint2 raw, float& a, float& b, float& c) {
    __half2 lo = *reinterpret_cast<__half2*>(&raw.x);
    __half2 hi = *reinterpret_cast<__half2*>(&raw.y);
    a = __low2float(lo);
    b = __high2float(lo);
    c = __low2float(hi);
}

// One voxel: owner crop (ox, oy, oz), own vector (v0, v1, v2) already loaded.  fp32 arithmetic op for op as the reference
// (vector_to_embedding.py:79-132, eval.py:274-276, skeleton.py:678-693).
template <typename L>
__device__ __forceinline__ int32_t follow_one(const uint2* __restrict__ vec4, const L* __restrict__ labels, const AssignGeom& g,
                                              const FollowParams& p, int x, int y, int z, int ox, int oy, int oz, float v0, float v1,
                                              float v2) {
    const int Zl = g.win_hi - g.win_lo;
    float mx = __fadd_rn((float)(x - ox), __fmul_rn(v0, p.sc[0][0]));
    float my = __fadd_rn((float)(y - oy), __fmul_rn(v1, p.sc[0][1]));
    float mz = __fadd_rn((float)(z - oz), __fmul_rn(v2, p.sc[0][2]));
    const int hd = p.h * p.d;
    for (int it = 1; it < p.n_iter; ++it) {
        int qx, qy, qz;
        long long j = flat_index(p, mx, my, mz, qx, qy, qz);
        // crop-local flat index -> crop-local coordinates (handles the clamp-to-k wrap)
        if (!(qx < p.w && qy < p.h && qz < p.d && j == ((long long)qx * p.h + qy) * p.d + qz)) {
            qx = (int)(j / hd);
            int r = (int)(j - (long long)qx * hd);
            qy = r / p.d;
            qz = r - qy * p.d;
        }
        long long gi = ((long long)(ox + qx) * g.Y + (oy + qy)) * Zl + (oz + qz - g.win_lo);
        float g0, g1, g2;
        unpack_vec4(vec4[gi], g0, g1, g2);
        if (g0 == 0.0f && g1 == 0.0f && g2 == 0.0f) break;  // fixed point: all later hops add 0
        mx = __fadd_rn(mx, __fmul_rn(g0, p.sc[it][0]));
        my = __fadd_rn(my, __fmul_rn(g1, p.sc[it][1]));
        mz = __fadd_rn(mz, __fmul_rn(g2, p.sc[it][2]));
    }
    // eval.py:274-276 (+ crop origin, fp32) then skeleton.py:678-693
    float ex = __fadd_rn(mx, (float)ox), ey = __fadd_rn(my, (float)oy), ez = __fadd_rn(mz, (float)oz);
    int xi = (int)clampf(rintf(ex), 0.0f, (float)(g.X - 1));
    int yi = (int)clampf(rintf(ey), 0.0f, (float)(g.Y - 1));
    int zi = (int)clampf(rintf(ez), 0.0f, (float)(g.Z - 1));
    return (int32_t)labels[((long long)xi * g.Y + yi) * g.Z + zi];
}

template <typename L>
__global__ void __launch_bounds__(256)
follow_assign_kernel(const uint2* __restrict__ vec4, const L* __restrict__ labels,
                     int32_t* __restrict__ out, AssignGeom g, const int32_t* __restrict__ own_x,
                     const int32_t* __restrict__ own_y, const int32_t* __restrict__ own_z,
                     FollowParams p) {
    const int zspan = g.z_hi - g.z_lo;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)g.X * g.Y * zspan;
    if (t >= total) return;
    int z = g.z_lo + (int)(t % zspan);
    long long xy = t / zspan;
    int y = (int)(xy % g.Y);
    int x = (int)(xy / g.Y);
    const int Zl = g.win_hi - g.win_lo;
    long long self = ((long long)x * g.Y + y) * Zl + (z - g.win_lo);

    int ox = own_x[x], oy = own_y[y], oz = own_z[z];
    const long long oidx = ((long long)x * g.Y + y) * zspan + (z - g.z_lo);  // out is (X, Y, z_hi-z_lo)
    if ((ox | oy | oz) < 0) {  // no crop interior covers this voxel (eval.py:245 zeros)
        out[oidx] = 0;
        return;
    }
    float v0, v1, v2;
    unpack_vec4(vec4[self], v0, v1, v2);
    out[oidx] = follow_one<L>(vec4, labels, g, p, x, y, z, ox, oy, oz, v0, v1, v2);
}

// The same, two z-neighbours per thread (round 3).  Most voxels of a real volume are background: their vector is zero,
// the follow stops at its first hop and the answer is the label at the voxel's own position.  For them the kernel is a
// stream -- one 16-byte load of two vectors, one 8-byte label load, one 8-byte store per thread, 32-bit index arithmetic
// (the one-voxel kernel spends its time in 64-bit divisions and 8-byte accesses: 1.3 TB/s of real traffic,
// profiles/r03_stage23_hbm_traffic_pmc.json) -- and only voxels with a vector take the dependent hops.  Needs even
// zspan / window / z offsets and int32 labels; blockIdx.y = x.
__global__ void __launch_bounds__(256)
follow_assign2_kernel(const uint4* __restrict__ vec4x2, const int32_t* __restrict__ labels, int32_t* __restrict__ out, AssignGeom g,
                      const int32_t* __restrict__ own_x, const int32_t* __restrict__ own_y, const int32_t* __restrict__ own_z,
                      FollowParams p, const bool tiled) {
    const int zspan = g.z_hi - g.z_lo, zh = zspan >> 1;
    const int x = blockIdx.y;
    int y, zp;
    if (tiled) {
        // a wave = 8 rows x 16 z voxels (8 lanes = one 128-byte line of vectors): objects are compact, so far fewer
        // waves contain a voxel that takes the dependent hops than with 128 consecutive z voxels of one row per wave
        // (the lattice blob field of bench.py: 82 % of the row-segment waves against about a quarter of these)
        const int lane = threadIdx.x & 63, wv = blockIdx.x * 4 + (threadIdx.x >> 6), zq = zh >> 3;
        const int ty = wv / zq;
        y = ty * 8 + (lane >> 3);
        zp = (wv - ty * zq) * 8 + (lane & 7);
        if (y >= g.Y) return;
    } else {
        const int t = blockIdx.x * 256 + threadIdx.x;   // pair index inside the x-plane
        if (t >= g.Y * zh) return;
        y = t / zh;
        zp = t - y * zh;
    }
    const int z = g.z_lo + 2 * zp;
    const int Zl = g.win_hi - g.win_lo;
    const long long row = (long long)x * g.Y + y;
    const long long self = row * Zl + (z - g.win_lo);                 // even
    const long long oidx = row * zspan + (z - g.z_lo);                // even
    const int ox = own_x[x], oy = own_y[y];
    const int oz0 = own_z[z], oz1 = own_z[z + 1];
    const uint4 raw = vec4x2[self >> 1];
    const uint2 r0 = make_uint2(raw.x, raw.y), r1 = make_uint2(raw.z, raw.w);
    // the label at the voxel's own position: the answer for a zero vector (embedding = position, skeleton.py:678-693)
    const int2 own = *reinterpret_cast<const int2*>(labels + (row * g.Z + z));
    int2 res;
    {
        float v0, v1, v2;
        unpack_vec4(r0, v0, v1, v2);
        if ((ox | oy | oz0) < 0)
            res.x = 0;
        else if (v0 == 0.0f && v1 == 0.0f && v2 == 0.0f)
            res.x = own.x;
        else
            res.x = follow_one<int32_t>(reinterpret_cast<const uint2*>(vec4x2), labels, g, p, x, y, z, ox, oy, oz0, v0, v1, v2);
    }
    {
        float v0, v1, v2;
        unpack_vec4(r1, v0, v1, v2);
        if ((ox | oy | oz1) < 0)
            res.y = 0;
        else if (v0 == 0.0f && v1 == 0.0f && v2 == 0.0f)
            res.y = own.y;
        else
            res.y = follow_one<int32_t>(reinterpret_cast<const uint2*>(vec4x2), labels, g, p, x, y, z + 1, ox, oy, oz1, v0, v1, v2);
    }
    *reinterpret_cast<int2*>(out + oidx) = res;
}

// ---------------------------------------------------------------- layout kernels
__global__ void __launch_bounds__(256) interleave_kernel(const __half* __restrict__ planar,
                                                         uint2* __restrict__ v4, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        __half2 lo = __halves2half2(planar[i], planar[i + n]);
        __half2 hi = __halves2half2(planar[i + 2 * n], __ushort_as_half(0));
        uint2 r;
        r.x = *reinterpret_cast<unsigned*>(&lo);
        r.y = *reinterpret_cast<unsigned*>(&hi);
        v4[i] = r;
    }
}

__global__ void __launch_bounds__(256) deinterleave_kernel(const uint2* __restrict__ v4,
                                                           __half* __restrict__ planar,
                                                           long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint2 r = v4[i];
        __half2 lo = *reinterpret_cast<__half2*>(&r.x);
        __half2 hi = *reinterpret_cast<__half2*>(&r.y);
        planar[i] = __low2half(lo);
        planar[i + n] = __high2half(lo);
        planar[i + 2 * n] = __low2half(hi);
    }
}

int fill_params(FollowParams& p, int w, int h, int d, const float* step_scale_host, int n_iter) {
    SK_CHECK_ARG(w > 0 && h > 0 && d > 0, "follow: crop extents must be positive (%d,%d,%d)", w, h, d);
    SK_CHECK_ARG(n_iter >= 1 && n_iter <= kMaxIter, "follow: N must be in [1,%d], got %d", kMaxIter,
                 n_iter);
    SK_CHECK_ARG(step_scale_host != nullptr, "follow: step_scale_host is NULL");
    p.w = w;
    p.h = h;
    p.d = d;
    p.n_iter = n_iter;
    p.fw = (float)w;
    p.fh = (float)h;
    p.fd = (float)d;
    p.nvox = (long long)w * h * d;
    p.fmax_flat = (float)(p.nvox - 1);
    for (int i = 0; i < n_iter; ++i)
        for (int c = 0; c < 3; ++c) p.sc[i][c] = step_scale_host[i * 3 + c];
    return SK_OK;
}

}  // namespace

extern "C" {

int sk_vec_interleave(const void* vec_planar, void* vec4, int64_t nvox, void* stream) {
    SK_CHECK_ARG(vec_planar && vec4 && nvox > 0, "sk_vec_interleave: bad arguments");
    interleave_kernel<<<sk::stream_grid(nvox, 256), 256, 0, (hipStream_t)stream>>>(
        (const __half*)vec_planar, (uint2*)vec4, nvox);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_vec_deinterleave(const void* vec4, void* vec_planar, int64_t nvox, void* stream) {
    SK_CHECK_ARG(vec_planar && vec4 && nvox > 0, "sk_vec_deinterleave: bad arguments");
    deinterleave_kernel<<<sk::stream_grid(nvox, 256), 256, 0, (hipStream_t)stream>>>(
        (const uint2*)vec4, (__half*)vec_planar, nvox);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_vector_to_embedding(const void* vec, int vec_dtype, float* embed, int w, int h, int d,
                           const float* step_scale_host, int n_iter, void* stream) {
    SK_CHECK_ARG(vec && embed, "sk_vector_to_embedding: NULL pointer");
    SK_CHECK_ARG(vec_dtype == SK_F16 || vec_dtype == SK_F32,
                 "sk_vector_to_embedding: vector dtype must be fp16 or fp32");
    FollowParams p;
    int rc = fill_params(p, w, h, d, step_scale_host, n_iter);
    if (rc) return rc;
    unsigned grid = sk::cdiv(p.nvox, 256);
    if (vec_dtype == SK_F16)
        vec2embed_kernel<__half><<<grid, 256, 0, (hipStream_t)stream>>>((const __half*)vec, embed, p);
    else
        vec2embed_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)vec, embed, p);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_index_skeleton_by_embed(const void* labels, int label_dtype, int lx, int ly, int lz,
                               const float* embed, int64_t n, int32_t* out, void* stream) {
    SK_CHECK_ARG(labels && embed && out, "sk_index_skeleton_by_embed: NULL pointer");
    SK_CHECK_ARG(lx > 0 && ly > 0 && lz > 0 && n > 0, "sk_index_skeleton_by_embed: bad extents");
    SK_CHECK_ARG(label_dtype == SK_I16 || label_dtype == SK_I32,
                 "sk_index_skeleton_by_embed: labels must be int16 or int32");
    unsigned grid = sk::cdiv(n, 256);
    if (label_dtype == SK_I16)
        index_by_embed_kernel<int16_t><<<grid, 256, 0, (hipStream_t)stream>>>(
            (const int16_t*)labels, lx, ly, lz, embed, n, out);
    else
        index_by_embed_kernel<int32_t><<<grid, 256, 0, (hipStream_t)stream>>>(
            (const int32_t*)labels, lx, ly, lz, embed, n, out);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_follow_assign(const void* vec4, const void* labels, int label_dtype, int32_t* out, int X,
                     int Y, int Z, int win_lo, int win_hi, const int32_t* owner_x,
                     const int32_t* owner_y, const int32_t* owner_z, int eff_w, int eff_h, int eff_d,
                     const float* step_scale_host, int n_iter, int z_lo, int z_hi, void* stream) {
    SK_CHECK_ARG(vec4 && labels && out && owner_x && owner_y && owner_z,
                 "sk_follow_assign: NULL pointer");
    SK_CHECK_ARG(X > 0 && Y > 0 && Z > 0, "sk_follow_assign: bad volume extents");
    SK_CHECK_ARG(0 <= win_lo && win_lo < win_hi && win_hi <= Z,
                 "sk_follow_assign: bad z window [%d,%d) for Z=%d", win_lo, win_hi, Z);
    SK_CHECK_ARG(win_lo <= z_lo && z_lo <= z_hi && z_hi <= win_hi,
                 "sk_follow_assign: z range [%d,%d) outside window [%d,%d)", z_lo, z_hi, win_lo, win_hi);
    SK_CHECK_ARG(eff_w <= X && eff_h <= Y && eff_d <= Z, "sk_follow_assign: crop exceeds volume");
    SK_CHECK_ARG(label_dtype == SK_I16 || label_dtype == SK_I32,
                 "sk_follow_assign: labels must be int16 or int32");
    if (z_lo == z_hi) return SK_OK;
    FollowParams p;
    int rc = fill_params(p, eff_w, eff_h, eff_d, step_scale_host, n_iter);
    if (rc) return rc;
    AssignGeom g{X, Y, Z, win_lo, win_hi, z_lo, z_hi};
    long long total = (long long)X * Y * (z_hi - z_lo);
    SK_CHECK_ARG(total / 256 < 0x7fffffffLL, "sk_follow_assign: volume too large for one launch");
    const int zspan = z_hi - z_lo;
    // The two-voxel kernel answers a zero-vector voxel with the label at its own position.  That equals the reference's
    // fp32 arithmetic (vector_to_embedding.py:109-130) only while the crop-local flat index is exact in fp32, i.e. for
    // crops of at most 2^24 voxels (the production crop: 12.5 M); above that the reference's rounded index may hop to a
    // voxel with a non-zero vector, which only the one-voxel kernel reproduces.
    if (label_dtype == SK_I32 && p.nvox <= (1LL << 24) && zspan % 2 == 0 && (win_hi - win_lo) % 2 == 0 && (z_lo - win_lo) % 2 == 0 && Z % 2 == 0 &&
        z_lo % 2 == 0 && X <= 65535 && ((uintptr_t)vec4 & 15) == 0 && ((uintptr_t)labels & 7) == 0 && ((uintptr_t)out & 7) == 0) {
        const bool tiled = zspan % 16 == 0;
        const long long waves = tiled ? (long long)((Y + 7) / 8) * (zspan / 16) : ((long long)Y * (zspan / 2) + 63) / 64;
        dim3 grid2(sk::cdiv(waves, 4), (unsigned)X);
        follow_assign2_kernel<<<grid2, 256, 0, (hipStream_t)stream>>>((const uint4*)vec4, (const int32_t*)labels, out, g, owner_x,
                                                                      owner_y, owner_z, p, tiled);
        SK_CHECK_LAUNCH();
        return SK_OK;
    }
    unsigned grid = sk::cdiv(total, 256);
    if (label_dtype == SK_I16)
        follow_assign_kernel<int16_t><<<grid, 256, 0, (hipStream_t)stream>>>(
            (const uint2*)vec4, (const int16_t*)labels, out, g, owner_x, owner_y, owner_z, p);
    else
        follow_assign_kernel<int32_t><<<grid, 256, 0, (hipStream_t)stream>>>(
            (const uint2*)vec4, (const int32_t*)labels, out, g, owner_x, owner_y, owner_z, p);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

}  // extern "C"
