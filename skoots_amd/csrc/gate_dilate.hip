// Stage-1 tail of skoots.lib.eval.eval(): gate, dilate, threshold, interior scatter.
//
// Replaces (reference file:line)
//   skoots/lib/eval.py:145-150      channel split + gating by prob > 0.8
//   skoots/lib/morphology.py:155-199 binary_dilation (3x3x3 max) + 2x binary_dilation_2d
//                                   (3x3x1 max), called at eval.py:152-157
//   skoots/lib/eval.py:160-176      interior crop scatter, skeleton > 0.8
//
// The reference materialises 3 x 27-channel one-hot conv3d outputs (194 MB fp32 per
// call).  Max filters commute with the final threshold and compose to one box of
// radius (3,3,1), zero padded at the tile faces, so the whole thing is a separable
// OR over a byte mask staged in LDS: HBM traffic is one read of channels 3,4 around
// the interior, one read of channels 0..2 on the interior, and the interior writes.
#include "common.h"

namespace {

constexpr int kMaxTiles = 16;

struct TileGeom {
    long long off;        // element offset of this tile's (channel 0, 0, 0, 0) in out5
    int ox, oy, oz;       // tile origin in the volume
    int lx, ly, lz;       // write box (tile-local, inside the interior): [l, h)
    int hx, hy, hz;
    int nbx, nblocks;     // patch grid of this tile
};

struct GateGeom {
    int w, h, d;          // tile extents
    long long sc, sx, sy; // element strides of out5: channel, x, y (z is contiguous)
    int X, Y, Z;          // volume extents
    int px, py;           // interior patch handled per block
    float prob_thr, skel_thr;
    int ntiles;
    TileGeom t[kMaxTiles];
};

template <typename T>
__device__ __forceinline__ float ld(const T* p, long long i);
template <>
__device__ __forceinline__ float ld<__half>(const __half* p, long long i) {
    return __half2float(p[i]);
}
template <>
__device__ __forceinline__ float ld<float>(const float* p, long long i) {
    return p[i];
}

constexpr int RX = 3, RY = 3;  // 1+1+1 in x,y (eval.py:152-157); z radius 1 is hard-wired in phase 2

template <typename T>
__global__ void __launch_bounds__(256)
gate_dilate_scatter_kernel(const T* __restrict__ out5_all, GateGeom g, uint2* __restrict__ vec4,
                           __half* __restrict__ vec_planar, uint8_t* __restrict__ skeleton) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const TileGeom tg = g.t[blockIdx.y];
    if ((int)blockIdx.x >= tg.nblocks) return;  // grid.x is sized for the largest write box
    const T* out5 = out5_all + tg.off;
    const int bx = blockIdx.x % tg.nbx, by = blockIdx.x / tg.nbx;
    const int x0 = tg.lx + bx * g.px, y0 = tg.ly + by * g.py;  // tile-local patch origin
    const int sx = g.px + 2 * RX, sy = g.py + 2 * RY;          // staged columns
    const int d = g.d;
    unsigned char* m0 = smem;                      // [sx][sy][d]
    unsigned char* m1 = smem + (size_t)sx * sy * d;  // [sx][sy][d]
    const long long plane = g.sc;
    const int tid = threadIdx.x, nth = blockDim.x;

    // phase 1: mask = skel > skel_thr && prob > prob_thr, zero outside the tile
    for (int i = tid; i < sx * sy * d; i += nth) {
        int z = i % d;
        int c = i / d;
        int ly = c % sy, lx = c / sy;
        int tx = x0 - RX + lx, ty = y0 - RY + ly;
        unsigned char m = 0;
        if (tx >= 0 && tx < g.w && ty >= 0 && ty < g.h) {
            long long o = (long long)tx * g.sx + (long long)ty * g.sy + z;
            float skel = ld(out5, 3 * plane + o);
            float prob = ld(out5, 4 * plane + o);
            m = (prob > g.prob_thr && skel > g.skel_thr) ? 1 : 0;
        }
        m0[i] = m;
    }
    __syncthreads();
    // phase 2: z dilation (radius 1), zero padded at the tile faces
    for (int i = tid; i < sx * sy * d; i += nth) {
        int z = i % d;
        unsigned char m = m0[i];
        if (z > 0) m |= m0[i - 1];
        if (z + 1 < d) m |= m0[i + 1];
        m1[i] = m;
    }
    __syncthreads();
    // phase 3: y dilation (radius 3) for the columns the x pass needs -> m0[sx][py][d]
    for (int i = tid; i < sx * g.py * d; i += nth) {
        int z = i % d;
        int c = i / d;
        int ly = c % g.py, lx = c / g.py;
        unsigned char m = 0;
#pragma unroll
        for (int k = 0; k <= 2 * RY; ++k) m |= m1[((size_t)lx * sy + ly + k) * d + z];
        m0[i] = m;
    }
    __syncthreads();
    // phase 4: x dilation (radius 3) + scatter of the interior
    const int iz0 = tg.lz, iz1 = tg.hz;
    const int idp = iz1 - iz0;
    for (int i = tid; i < g.px * g.py * idp; i += nth) {
        int zz = i % idp;
        int c = i / idp;
        int ly = c % g.py, lx = c / g.py;
        int tx = x0 + lx, ty = y0 + ly, tz = iz0 + zz;
        if (tx >= tg.hx || ty >= tg.hy) continue;
        unsigned char m = 0;
#pragma unroll
        for (int k = 0; k <= 2 * RX; ++k) m |= m0[((size_t)(lx + k) * g.py + ly) * d + tz];
        long long vo = ((long long)(tg.ox + tx) * g.Y + (tg.oy + ty)) * g.Z + (tg.oz + tz);
        skeleton[vo] = m;
        // vectors: vec * (prob > thr), stored fp16 (eval.py:149,175)
        long long o = (long long)tx * g.sx + (long long)ty * g.sy + tz;
        bool gate = ld(out5, 4 * plane + o) > g.prob_thr;
        float v0 = ld(out5, o), v1 = ld(out5, plane + o), v2 = ld(out5, 2 * plane + o);
        if (!gate) {
            v0 = copysignf(0.0f, v0);
            v1 = copysignf(0.0f, v1);
            v2 = copysignf(0.0f, v2);
        }
        __half h0 = __float2half_rn(v0), h1 = __float2half_rn(v1), h2 = __float2half_rn(v2);
        if (vec4) {
            __half2 lo = __halves2half2(h0, h1);
            __half2 hi = __halves2half2(h2, __ushort_as_half(0));
            uint2 r;
            r.x = *reinterpret_cast<unsigned*>(&lo);
            r.y = *reinterpret_cast<unsigned*>(&hi);
            vec4[vo] = r;
        }
        if (vec_planar) {
            long long nv = (long long)g.X * g.Y * g.Z;
            vec_planar[vo] = h0;
            vec_planar[vo + nv] = h1;
            vec_planar[vo + 2 * nv] = h2;
        }
    }
}

// Library max filter (binary_dilation / binary_dilation_2d on arbitrary fp32 maps).
__global__ void __launch_bounds__(256) max_filter_kernel(const float* __restrict__ in,
                                                         float* __restrict__ out, int w, int h,
                                                         int d, int rx, int ry, int rz) {
    long long n = (long long)w * h * d;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int z = (int)(i % d);
    long long c = i / d;
    int y = (int)(c % h), x = (int)(c / h);
    // zero padding: out-of-bounds taps contribute 0.0 (morphology.py:170-175)
    bool clipped = (x - rx < 0) || (x + rx >= w) || (y - ry < 0) || (y + ry >= h) || (z - rz < 0) ||
                   (z + rz >= d);
    float m = clipped ? 0.0f : -INFINITY;
    bool any = clipped;
    for (int dx = -rx; dx <= rx; ++dx) {
        int xx = x + dx;
        if (xx < 0 || xx >= w) continue;
        for (int dy = -ry; dy <= ry; ++dy) {
            int yy = y + dy;
            if (yy < 0 || yy >= h) continue;
            for (int dz = -rz; dz <= rz; ++dz) {
                int zz = z + dz;
                if (zz < 0 || zz >= d) continue;
                float v = in[((long long)xx * h + yy) * d + zz];
                m = any ? fmaxf(m, v) : v;
                any = true;
            }
        }
    }
    out[i] = m;
}

}  // namespace

extern "C" {

int sk_gate_dilate_scatter(const void* out5, int out_dtype, int n_tiles, const int64_t* tile_offsets_host,
                           int64_t stride_c, int64_t stride_x, int64_t stride_y, int w, int h, int d,
                           const int* origins_host, const int* box_lo_host, const int* box_hi_host,
                           void* vec4, void* vec_planar, uint8_t* skeleton, int X, int Y, int Z,
                           float prob_thr, float skel_thr, void* stream) {
    SK_CHECK_ARG(out5 && skeleton && tile_offsets_host && origins_host && box_lo_host && box_hi_host,
                 "sk_gate_dilate_scatter: NULL pointer");
    SK_CHECK_ARG(out_dtype == SK_F16 || out_dtype == SK_F32,
                 "sk_gate_dilate_scatter: out dtype must be fp16 or fp32");
    SK_CHECK_ARG(n_tiles >= 1 && n_tiles <= kMaxTiles, "sk_gate_dilate_scatter: 1..%d tiles per call", kMaxTiles);
    SK_CHECK_ARG(w > 0 && h > 0 && d > 0 && stride_y >= d && stride_x >= stride_y && stride_c > 0,
                 "sk_gate_dilate_scatter: bad tile extents / strides");
    GateGeom g{};
    g.w = w;
    g.h = h;
    g.d = d;
    g.sc = stride_c;
    g.sx = stride_x;
    g.sy = stride_y;
    g.X = X;
    g.Y = Y;
    g.Z = Z;
    g.px = g.py = 16;
    g.prob_thr = prob_thr;
    g.skel_thr = skel_thr;
    g.ntiles = n_tiles;
    size_t lds = 2ull * (g.px + 2 * RX) * (g.py + 2 * RY) * d;
    if (lds > 64 * 1024) {
        g.px = g.py = 8;
        lds = 2ull * (g.px + 2 * RX) * (g.py + 2 * RY) * d;
    }
    SK_CHECK_ARG(lds <= 96 * 1024, "sk_gate_dilate_scatter: tile depth %d too large", d);
    int max_blocks = 0;
    for (int i = 0; i < n_tiles; ++i) {
        const int* lo = box_lo_host + 3 * i;
        const int* hi = box_hi_host + 3 * i;
        const int* org = origins_host + 3 * i;
        SK_CHECK_ARG(0 <= lo[0] && lo[0] < hi[0] && hi[0] <= w && 0 <= lo[1] && lo[1] < hi[1] && hi[1] <= h &&
                         0 <= lo[2] && lo[2] < hi[2] && hi[2] <= d,
                     "sk_gate_dilate_scatter: write box [%d:%d,%d:%d,%d:%d) must be a non-empty box inside the "
                     "tile (%d,%d,%d)", lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], w, h, d);
        SK_CHECK_ARG(org[0] >= 0 && org[1] >= 0 && org[2] >= 0 && org[0] + w <= X && org[1] + h <= Y &&
                         org[2] + d <= Z,
                     "sk_gate_dilate_scatter: tile [%d+%d,%d+%d,%d+%d) outside volume (%d,%d,%d)", org[0], w,
                     org[1], h, org[2], d, X, Y, Z);
        TileGeom& t = g.t[i];
        t.off = tile_offsets_host[i];
        t.ox = org[0];
        t.oy = org[1];
        t.oz = org[2];
        t.lx = lo[0];
        t.ly = lo[1];
        t.lz = lo[2];
        t.hx = hi[0];
        t.hy = hi[1];
        t.hz = hi[2];
        t.nbx = (hi[0] - lo[0] + g.px - 1) / g.px;
        t.nblocks = t.nbx * ((hi[1] - lo[1] + g.py - 1) / g.py);
        if (t.nblocks > max_blocks) max_blocks = t.nblocks;
    }
    dim3 grid(max_blocks, n_tiles);
    if (out_dtype == SK_F16) {
        if (lds > 48 * 1024)
            SK_CHECK_HIP(hipFuncSetAttribute((const void*)gate_dilate_scatter_kernel<__half>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        gate_dilate_scatter_kernel<__half><<<grid, 256, lds, (hipStream_t)stream>>>(
            (const __half*)out5, g, (uint2*)vec4, (__half*)vec_planar, skeleton);
    } else {
        if (lds > 48 * 1024)
            SK_CHECK_HIP(hipFuncSetAttribute((const void*)gate_dilate_scatter_kernel<float>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        gate_dilate_scatter_kernel<float><<<grid, 256, lds, (hipStream_t)stream>>>(
            (const float*)out5, g, (uint2*)vec4, (__half*)vec_planar, skeleton);
    }
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_max_filter3d(const float* in, float* out, int w, int h, int d, int rx, int ry, int rz,
                    void* stream) {
    SK_CHECK_ARG(in && out && in != out, "sk_max_filter3d: bad pointers");
    SK_CHECK_ARG(w > 0 && h > 0 && d > 0 && rx >= 0 && ry >= 0 && rz >= 0,
                 "sk_max_filter3d: bad extents");
    long long n = (long long)w * h * d;
    max_filter_kernel<<<sk::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(in, out, w, h, d, rx, ry, rz);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

}  // extern "C"
