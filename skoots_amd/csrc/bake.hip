// Training-target baking: for every voxel of instance k the coordinates of the nearest point of
// skeleton k under an anisotropic Euclidean distance, and the 3x3x3 "mean of the non-empty neighbours" smoothing.
//
// Replaces skoots/lib/skeleton.py:448-528 (bake_skeleton; the reference's GPU path is a Triton kernel,
// skeleton.py:51-251,258-367, its CPU path torch.cdist + argmin, skeleton.py:370-445) and skeleton.py:18-48
// (average_baked_skeletons).  Distances are formed exactly (integer coordinates, double arithmetic) and the first
// minimal skeleton point wins -- the rule of oracle/bake.py; the reference's own winner among equidistant points
// depends on cdist's rounding.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) bake_kernel(const int* __restrict__ masks, const int* __restrict__ ids,
                                                   const int* __restrict__ offsets, const float* __restrict__ points,
                                                   int K, int X, int Y, int Z, double ax, double ay, double az,
                                                   float* __restrict__ baked, float* __restrict__ distance) {
    const long long n = (long long)X * Y * Z;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int id = masks[i];
        float bx = 0.0f, by = 0.0f, bz = 0.0f, dist = 0.0f;
        if (id != 0) {
            int lo = 0, hi = K - 1, k = -1;  // ids are sorted
            while (lo <= hi) {
                const int mid = (lo + hi) >> 1;
                const int v = ids[mid];
                if (v == id) {
                    k = mid;
                    break;
                }
                if (v < id)
                    lo = mid + 1;
                else
                    hi = mid - 1;
            }
            if (k >= 0) {
                const int z = (int)(i % Z);
                const long long t = i / Z;
                const int y = (int)(t % Y), x = (int)(t / Y);
                double best = 1.0e300;
                for (int p = offsets[k]; p < offsets[k + 1]; ++p) {
                    const float px = points[3 * p], py = points[3 * p + 1], pz = points[3 * p + 2];
                    const double dx = ((double)x - px) * ax, dy = ((double)y - py) * ay, dz = ((double)z - pz) * az;
                    const double d2 = dx * dx + dy * dy + dz * dz;
                    if (d2 < best) {  // strict: the first minimal point wins
                        best = d2;
                        bx = px;
                        by = py;
                        bz = pz;
                    }
                }
                dist = (float)sqrt(best);
            }
        }
        baked[i] = bx;
        baked[n + i] = by;
        baked[2 * n + i] = bz;
        if (distance) distance[i] = dist;
    }
}

__global__ void __launch_bounds__(256) average_baked_kernel(const float* __restrict__ in, float* __restrict__ out, int C,
                                                            int X, int Y, int Z) {
    const long long nv = (long long)X * Y * Z, n = nv * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long v = i % nv;
        const float* ch = in + (i - v);
        const int z = (int)(v % Z);
        const long long t = v / Z;
        const int y = (int)(t % Y), x = (int)(t / Y);
        double s = 0.0;
        int cnt = 0;
        for (int dx = -1; dx <= 1; ++dx)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dz = -1; dz <= 1; ++dz) {
                    const int xx = x + dx, yy = y + dy, zz = z + dz;
                    if (xx < 0 || xx >= X || yy < 0 || yy >= Y || zz < 0 || zz >= Z) continue;
                    const float w = ch[((long long)xx * Y + yy) * Z + zz];
                    s += (double)w;
                    cnt += w > 0.0f;   // a coordinate of exactly 0 counts as empty (the reference's rule)
                }
        out[i] = (float)(s / (double)(cnt ? cnt : 1));
    }
}

}  // namespace

extern "C" {

int sk_bake_skeleton(const int32_t* masks, const int32_t* ids, const int32_t* offsets, const float* points, int K, int X,
                     int Y, int Z, const float* anisotropy_host, float* baked, float* distance, void* stream) {
    SK_CHECK_ARG(masks && baked && anisotropy_host && X > 0 && Y > 0 && Z > 0 && K >= 0, "sk_bake_skeleton: bad arguments");
    SK_CHECK_ARG(K == 0 || (ids && offsets && points), "sk_bake_skeleton: NULL skeleton table");
    const long long n = (long long)X * Y * Z;
    bake_kernel<<<sk::stream_grid(n, 256, 2), 256, 0, (hipStream_t)stream>>>(masks, ids, offsets, points, K, X, Y, Z,
                                                                             (double)anisotropy_host[0], (double)anisotropy_host[1],
                                                                             (double)anisotropy_host[2], baked, distance);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_average_baked_skeletons(const float* baked, float* out, int C, int X, int Y, int Z, void* stream) {
    SK_CHECK_ARG(baked && out && baked != out && C > 0 && X > 0 && Y > 0 && Z > 0, "sk_average_baked_skeletons: bad arguments");
    const long long n = (long long)C * X * Y * Z;
    average_baked_kernel<<<sk::stream_grid(n, 256, 2), 256, 0, (hipStream_t)stream>>>(baked, out, C, X, Y, Z);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

}  // extern "C"
