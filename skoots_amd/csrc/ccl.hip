// Stage 2 of skoots.lib.eval.eval(): skeleton connected-component labelling.
//
// Replaces (reference file:line)
//   skoots/lib/flood_fill.py:125-140   flood_all  (scipy.ndimage.label + id offset)
//   skoots/lib/flood_fill.py:237-261   get_adjacent_labels (seam scan)
//   skoots/lib/flood_fill.py:143-174   connected_components / dfs   (host, C++)
//   skoots/lib/flood_fill.py:177-234   replace / _in_place_replace  (LUT relabel)
//
// Per flood crop: union-find over the 6-neighbourhood with atomicMin on roots, so a
// component's root is its first voxel in raster order; roots are ranked with a
// prefix sum, which reproduces scipy's numbering (components numbered by first
// voxel in C order).  HBM-bound integer work, 4-byte parent per voxel.
#include <algorithm>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace {

struct Box {
    int X, Y, Z;        // volume
    int x0, y0, z0;     // crop origin
    int w, h, d;        // crop extents
};

constexpr int kScanChunk = 2048;  // elements per block in the rank passes

__device__ __forceinline__ long long vox(const Box& b, int i) {
    int hd = b.h * b.d;
    int x = i / hd;
    int r = i - x * hd;
    int y = r / b.d;
    int z = r - y * b.d;
    return ((long long)(b.x0 + x) * b.Y + (b.y0 + y)) * b.Z + (b.z0 + z);
}

__device__ __forceinline__ int find_root(const int* parent, int i) {
    int p = parent[i];
    while (p != i) {
        i = p;
        p = parent[i];
    }
    return i;
}

__device__ __forceinline__ void unite(int* parent, int a, int b) {
    // lock-free union by minimum index
    while (true) {
        a = find_root(parent, a);
        b = find_root(parent, b);
        if (a == b) return;
        if (a < b) {
            int t = a;
            a = b;
            b = t;
        }  // a > b: hang a under b
        int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;
    }
}

// init: parent = self for foreground, -1 for background; link along z immediately
__global__ void __launch_bounds__(256) ccl_init_kernel(const uint8_t* __restrict__ src,
                                                       int* __restrict__ parent, Box b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    parent[i] = src[vox(b, i)] ? i : -1;
}

__global__ void __launch_bounds__(256) ccl_merge_kernel(int* __restrict__ parent, Box b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (parent[i] < 0) return;
    int hd = b.h * b.d;
    int x = i / hd;
    int r = i - x * hd;
    int y = r / b.d;
    int z = r - y * b.d;
    if (z > 0 && parent[i - 1] >= 0) unite(parent, i, i - 1);
    if (y > 0 && parent[i - b.d] >= 0) unite(parent, i, i - b.d);
    if (x > 0 && parent[i - hd] >= 0) unite(parent, i, i - hd);
}

__global__ void __launch_bounds__(256) ccl_compress_kernel(int* __restrict__ parent, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (parent[i] < 0) return;
    // No other thread changes a root's parent in this kernel, and every chain ends at
    // a root, so reading un-compressed ancestors is safe.
    int r = find_root(parent, i);
    if (r != i) parent[i] = r;
}

// per-chunk root counts
__global__ void __launch_bounds__(256) ccl_count_kernel(const int* __restrict__ parent, int n,
                                                        int* __restrict__ chunk_count) {
    __shared__ int wsum[4];
    int base = blockIdx.x * kScanChunk;
    int c = 0;
    for (int k = threadIdx.x; k < kScanChunk; k += 256) {
        int i = base + k;
        if (i < n && parent[i] == i) ++c;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// single block: exclusive scan of chunk counts + crop bookkeeping
// state[0] running id, [1] comps in crop, [2] total comps, [3] id offset used by this crop
__global__ void __launch_bounds__(1024) ccl_scan_kernel(int* __restrict__ chunk_count, int nchunks,
                                                        int* __restrict__ state) {
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nchunks; base += 1024) {
        int i = base + threadIdx.x;
        int v = (i < nchunks) ? chunk_count[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            int t = (threadIdx.x >= o) ? part[threadIdx.x - o] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        int incl = part[threadIdx.x];
        if (i < nchunks) chunk_count[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int count = carry;
        int id = state[0] + 1;             // flood_fill.py:45-47  flood_all(crop, max_id + 1)
        state[3] = id;
        state[1] = count;
        state[2] += count;
        state[0] = count > 0 ? id + count : 0;  // flood_fill.py:138-140  mask.max()
    }
}

// rank roots: parent[root] = -(label) - 2  (label = rank + id, rank from 1)
__global__ void __launch_bounds__(256) ccl_rank_kernel(int* __restrict__ parent, int n,
                                                       const int* __restrict__ chunk_offset,
                                                       const int* __restrict__ state) {
    __shared__ int wpre[4];
    __shared__ int run;
    const int id = state[3];
    int base = blockIdx.x * kScanChunk;
    if (threadIdx.x == 0) run = chunk_offset[blockIdx.x];
    __syncthreads();
    // 8 rounds of 256 consecutive elements keep raster order
    for (int k = 0; k < kScanChunk; k += 256) {
        int i = base + k + threadIdx.x;
        bool root = (i < n) && (parent[i] == i);
        unsigned long long bal = __ballot(root);
        int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        int within = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wpre[wv] = __popcll(bal);
        __syncthreads();
        int before = 0;
        for (int q = 0; q < wv; ++q) before += wpre[q];
        int tot = wpre[0] + wpre[1] + wpre[2] + wpre[3];
        int r0 = run;
        if (root) parent[i] = -(r0 + before + within + 1 + id) - 2;
        __syncthreads();
        if (threadIdx.x == 0) run = r0 + tot;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) ccl_write_kernel(const int* __restrict__ parent,
                                                        int32_t* __restrict__ labels, Box b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int p = parent[i];
    int lab = 0;
    if (p != -1) {
        if (p >= 0) p = parent[p];  // compressed: p is the root, whose slot holds the code
        lab = -(p + 2);
    }
    labels[vox(b, i)] = lab;
}

// ------------------------------------------------------------------------------------------------------------------
// Fast path (round 3): the same union-find driven by the MASK.  A skeleton is a fraction of a percent foreground, and
// the kernels above move 29 B per voxel through seven one-thread-per-voxel passes, most of it 4-byte parent entries of
// background voxels (PMC: 8.1 GB per 1024x1024x256 pass for 2.4 GB of algorithmic traffic, profiles/r03_stage23_*).
// Here a thread owns 16 consecutive z voxels (one 16-byte load of the mask), the parent array is touched at FOREGROUND
// voxels only (background entries are never written nor read), z links inside a 16-voxel chunk are made while
// initialising (parent = start of the voxel's run), and compress + count share a pass: five data passes, each reading
// 1 B per voxel, plus the dense 4 B label write.  Needs d, Z and z0 to be multiples of 16 (16-byte aligned rows); other
// crops take the kernels above.  Same result bit for bit: a component's root is its first voxel in raster order
// whatever the order of the unions.
__device__ __forceinline__ uint4 mask16(const uint8_t* __restrict__ src, const Box& b, int i) {
    return *reinterpret_cast<const uint4*>(src + vox(b, i));
}
__device__ __forceinline__ unsigned fg_bits(const uint4 m) {   // bit k: byte k of the chunk is foreground
    unsigned r = 0;
    const unsigned w[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) r |= (((w[q] >> (8 * k)) & 0xffu) ? 1u : 0u) << (4 * q + k);
    return r;
}

__global__ void __launch_bounds__(256) ccl16_init_kernel(const uint8_t* __restrict__ src, int* __restrict__ parent, Box b, int nchunk16) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunk16) return;
    const int i0 = c * 16;
    const unsigned f = fg_bits(mask16(src, b, i0));
    if (!f) return;
    int start = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (!((f >> k) & 1u)) continue;
        if (k == 0 || !((f >> (k - 1)) & 1u)) start = i0 + k;   // a run begins: its first voxel is the run's root
        parent[i0 + k] = start;
    }
}

__global__ void __launch_bounds__(256) ccl16_merge_kernel(const uint8_t* __restrict__ src, int* __restrict__ parent, Box b, int nchunk16) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunk16) return;
    const int i0 = c * 16;
    const unsigned f = fg_bits(mask16(src, b, i0));
    if (!f) return;
    const int hd = b.h * b.d;
    const int x = i0 / hd;
    const int r = i0 - x * hd;
    const int y = r / b.d;
    const int z = r - y * b.d;
    // z: the chunk's first voxel against the last voxel of the chunk before it (links inside a chunk exist already)
    if ((f & 1u) && z > 0 && src[vox(b, i0 - 1)]) unite(parent, i0, i0 - 1);
    // y and x: one union per pair of overlapping runs (first voxel of every stretch where both are foreground)
    if (y > 0) {
        const unsigned both = f & fg_bits(mask16(src, b, i0 - b.d));
        unsigned first = both & ~(both << 1);
        while (first) {
            const int k = __ffs(first) - 1;
            first &= first - 1;
            unite(parent, i0 + k, i0 + k - b.d);
        }
    }
    if (x > 0) {
        const unsigned both = f & fg_bits(mask16(src, b, i0 - hd));
        unsigned first = both & ~(both << 1);
        while (first) {
            const int k = __ffs(first) - 1;
            first &= first - 1;
            unite(parent, i0 + k, i0 + k - hd);
        }
    }
}

// compress + per-block root count (blocks of kScanChunk voxels = 128 threads of 16 voxels, raster order)
__global__ void __launch_bounds__(128) ccl16_compress_count_kernel(const uint8_t* __restrict__ src, int* __restrict__ parent, Box b,
                                                                   int nchunk16, int* __restrict__ chunk_count) {
    __shared__ int wsum[2];
    const int c = blockIdx.x * 128 + threadIdx.x;
    int roots = 0;
    if (c < nchunk16) {
        const int i0 = c * 16;
        unsigned f = fg_bits(mask16(src, b, i0));
        while (f) {
            const int k = __ffs(f) - 1;
            f &= f - 1;
            const int i = i0 + k;
            const int r = find_root(parent, i);   // roots do not change in this kernel: reading uncompressed ancestors is safe
            if (r != i)
                parent[i] = r;
            else
                ++roots;
        }
    }
    for (int o = 32; o > 0; o >>= 1) roots += __shfl_down(roots, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = roots;
    __syncthreads();
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = wsum[0] + wsum[1];
}

// rank the roots of a block in raster order: parent[root] = -(label) - 2, label = rank + id (rank from 1)
__global__ void __launch_bounds__(128) ccl16_rank_kernel(const uint8_t* __restrict__ src, int* __restrict__ parent, Box b, int nchunk16,
                                                         const int* __restrict__ chunk_offset, const int* __restrict__ state) {
    __shared__ int wtot[2];
    const int id = state[3];
    const int c = blockIdx.x * 128 + threadIdx.x;
    unsigned rootbits = 0;
    int i0 = 0;
    if (c < nchunk16) {
        i0 = c * 16;
        unsigned f = fg_bits(mask16(src, b, i0));
        while (f) {
            const int k = __ffs(f) - 1;
            f &= f - 1;
            if (parent[i0 + k] == i0 + k) rootbits |= 1u << k;
        }
    }
    const int mine = __popc(rootbits);
    // exclusive prefix over the 128 threads (raster order = thread order)
    int incl = mine;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    int before = chunk_offset[blockIdx.x] + (wv ? wtot[0] : 0) + incl - mine;
    while (rootbits) {
        const int k = __ffs(rootbits) - 1;
        rootbits &= rootbits - 1;
        ++before;
        parent[i0 + k] = -(before + id) - 2;
    }
}

__global__ void __launch_bounds__(256) ccl16_write_kernel(const uint8_t* __restrict__ src, const int* __restrict__ parent,
                                                          int32_t* __restrict__ labels, Box b, int nchunk16) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunk16) return;
    const int i0 = c * 16;
    const unsigned f = fg_bits(mask16(src, b, i0));
    int lab[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        lab[k] = 0;
        if ((f >> k) & 1u) {
            int p = parent[i0 + k];
            if (p >= 0) p = parent[p];   // compressed: p is the root, whose slot holds the code
            lab[k] = -(p + 2);
        }
    }
    int4* out = reinterpret_cast<int4*>(labels + vox(b, i0));
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = make_int4(lab[4 * q], lab[4 * q + 1], lab[4 * q + 2], lab[4 * q + 3]);
}

__global__ void __launch_bounds__(256) seam_pairs_kernel(const int32_t* __restrict__ labels, int X,
                                                         int Y, int Z, int axis, int v,
                                                         int32_t* __restrict__ pairs,
                                                         int32_t* __restrict__ count, int capacity) {
    // plane coordinates (u, t) with t fastest in memory
    int nu, nt;
    long long su, st, base0, base1;
    if (axis == 0) {
        nu = Y; nt = Z; su = Z; st = 1;
        base0 = (long long)v * Y * Z; base1 = (long long)(v - 1) * Y * Z;
    } else if (axis == 1) {
        nu = X; nt = Z; su = (long long)Y * Z; st = 1;
        base0 = (long long)v * Z; base1 = (long long)(v - 1) * Z;
    } else {
        nu = X; nt = Y; su = (long long)Y * Z; st = Z;
        base0 = v; base1 = v - 1;
    }
    long long n = (long long)nu * nt;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int t = (int)(i % nt);
    long long u = i / nt;
    long long o = u * su + (long long)t * st;
    int a = labels[base0 + o], b = labels[base1 + o];
    if (a == 0 || b == 0) return;
    if (t > 0) {  // cheap de-duplication of runs
        int pa = labels[base0 + o - st], pb = labels[base1 + o - st];
        if (pa == a && pb == b) return;
    }
    int slot = atomicAdd(count, 1);
    if (slot < capacity) {
        pairs[2 * slot] = a;
        pairs[2 * slot + 1] = b;
    }
}

__global__ void __launch_bounds__(256) relabel_lut_kernel(int32_t* __restrict__ labels, long long n,
                                                          const int32_t* __restrict__ lut,
                                                          int lut_size) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int v = labels[i];
        if (v > 0 && v < lut_size) {
            int r = lut[v];
            if (r != v) labels[i] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Z-sharded labelling without a host round trip between its collectives (skoots_amd/lib/flood_fill.py: label_slab).
// ------------------------------------------------------------------------------------------------------------------
// positions (flat index into `labels`) of the non-zero entries, in any order; *count keeps counting past `capacity`
__global__ void __launch_bounds__(256) compact_nonzero_kernel(const int32_t* __restrict__ labels, long long n, long long* __restrict__ out,
                                                              unsigned long long* __restrict__ count, long long capacity) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    for (; i - lane < n; i += stride) {   // whole waves iterate together (ballot)
        const bool nz = i < n && labels[i] != 0;
        const unsigned long long bal = __ballot(nz);
        if (bal == 0) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(count, (unsigned long long)__popcll(bal));
        base = __shfl(base, 0);
        if (nz) {
            const long long slot = (long long)base + __popcll(bal & ((1ull << lane) - 1ull));
            if (slot < capacity) out[slot] = i;
        }
    }
}

// Union of label ids over seam pairs, on the device: one workgroup.  pairs (R, cap, 2) in rank-LOCAL ids, row r holds
// min(npairs[r], cap) valid pairs (upper rank's id, this rank's id): column 0 is shifted by offsets[min(r + 1, R - 1)],
// column 1 by offsets[r] (exclusive prefix sums of the ranks' component counts) to global ids.  lut (>= total + 1
// entries, pre-filled with the identity by the caller) ends with lut[id] = smallest id of id's component -- the
// partition of flood_fill.py:82-117; which member names a component is free here (the sharded path renumbers anyway).
__global__ void __launch_bounds__(1024) seam_union_kernel(const int32_t* __restrict__ meta, int R, int row_stride, int cap,
                                                          const long long* __restrict__ offsets, int32_t* __restrict__ lut,
                                                          long long lut_size) {
    for (int pass = 0; pass < 2; ++pass) {
        for (int r = 0; r < R; ++r) {
            const int32_t* row = meta + (long long)r * row_stride;
            const int n = min(row[1], cap);
            const long long o0 = offsets[min(r + 1, R - 1)], o1 = offsets[r];
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const long long ga = row[4 + 2 * i] + o0, gb = row[4 + 2 * i + 1] + o1;
                if (ga <= 0 || gb <= 0 || ga >= lut_size || gb >= lut_size) continue;
                if (pass == 0) {
                    unite(lut, (int)ga, (int)gb);
                } else {   // every id that appears in a pair points at its root
                    lut[ga] = find_root(lut, (int)ga);
                    lut[gb] = find_root(lut, (int)gb);
                }
            }
        }
        __syncthreads();
    }
}

// labels[i] = lut[labels[i] + *offset] for labels[i] > 0 (local ids -> merged global ids)
__global__ void __launch_bounds__(256) relabel_lut_offset_kernel(int32_t* __restrict__ labels, long long n, const int32_t* __restrict__ lut,
                                                                 long long lut_size, const long long* __restrict__ offset) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long off = *offset;
    for (; i < n; i += stride) {
        const int v = labels[i];
        if (v > 0) {
            const long long g = v + off;
            labels[i] = g < lut_size ? lut[g] : (int32_t)g;
        }
    }
}

}  // namespace

extern "C" {

int sk_compact_nonzero(const int32_t* labels, int64_t n, int64_t* positions, uint64_t* count, int64_t capacity, void* stream) {
    SK_CHECK_ARG(labels && positions && count && n > 0 && capacity > 0, "sk_compact_nonzero: bad arguments");
    compact_nonzero_kernel<<<sk::stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(labels, n, (long long*)positions,
                                                                                     (unsigned long long*)count, capacity);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_seam_union(const int32_t* meta, int ranks, int row_stride, int pair_capacity, const int64_t* offsets, int32_t* lut,
                  int64_t lut_size, void* stream) {
    SK_CHECK_ARG(meta && offsets && lut && ranks > 0 && pair_capacity >= 0 && row_stride >= 4 + 2 * pair_capacity && lut_size > 0,
                 "sk_seam_union: bad arguments");
    seam_union_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(meta, ranks, row_stride, pair_capacity, (const long long*)offsets, lut,
                                                           lut_size);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_relabel_lut_offset(int32_t* labels, int64_t n, const int32_t* lut, int64_t lut_size, const int64_t* offset, void* stream) {
    SK_CHECK_ARG(labels && lut && offset && n > 0 && lut_size > 0, "sk_relabel_lut_offset: bad arguments");
    relabel_lut_offset_kernel<<<sk::stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(labels, n, lut, lut_size, (const long long*)offset);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

size_t sk_ccl_workspace_bytes(int64_t n) {
    size_t chunks = (size_t)((n + kScanChunk - 1) / kScanChunk);
    return (size_t)n * sizeof(int) + (chunks + 16) * sizeof(int);
}

int sk_ccl_crop(const uint8_t* src, int32_t* labels, int X, int Y, int Z, int x0, int y0, int z0,
                int w, int h, int d, void* workspace, size_t workspace_bytes, int32_t* state,
                void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SK_CHECK_ARG(src && labels && workspace && state, "sk_ccl_crop: NULL pointer");
    SK_CHECK_ARG(w > 0 && h > 0 && d > 0 && x0 >= 0 && y0 >= 0 && z0 >= 0 && x0 + w <= X &&
                     y0 + h <= Y && z0 + d <= Z,
                 "sk_ccl_crop: crop [%d+%d,%d+%d,%d+%d) outside volume (%d,%d,%d)", x0, w, y0, h, z0,
                 d, X, Y, Z);
    long long n64 = (long long)w * h * d;
    SK_CHECK_ARG(n64 < 0x7fffffffLL - kScanChunk, "sk_ccl_crop: crop of %lld voxels is too large", n64);
    SK_CHECK_ARG(workspace_bytes >= sk_ccl_workspace_bytes(n64),
                 "sk_ccl_crop: workspace too small (%zu < %zu)", workspace_bytes,
                 sk_ccl_workspace_bytes(n64));
    int n = (int)n64;
    Box b{X, Y, Z, x0, y0, z0, w, h, d};
    int* parent = (int*)workspace;
    int* chunk = parent + n;
    int nchunks = (n + kScanChunk - 1) / kScanChunk;
    if (d % 16 == 0 && Z % 16 == 0 && z0 % 16 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)labels & 15) == 0) {
        // mask-driven path: 16 voxels per thread, the parent array touched at foreground voxels only
        const int nchunk16 = n / 16;
        const unsigned g16 = sk::cdiv(nchunk16, 256);
        ccl16_init_kernel<<<g16, 256, 0, stream>>>(src, parent, b, nchunk16);
        ccl16_merge_kernel<<<g16, 256, 0, stream>>>(src, parent, b, nchunk16);
        ccl16_compress_count_kernel<<<nchunks, 128, 0, stream>>>(src, parent, b, nchunk16, chunk);
        ccl_scan_kernel<<<1, 1024, 0, stream>>>(chunk, nchunks, state);
        ccl16_rank_kernel<<<nchunks, 128, 0, stream>>>(src, parent, b, nchunk16, chunk, state);
        ccl16_write_kernel<<<g16, 256, 0, stream>>>(src, parent, labels, b, nchunk16);
        SK_CHECK_LAUNCH();
        return SK_OK;
    }
    unsigned grid = sk::cdiv(n, 256);
    ccl_init_kernel<<<grid, 256, 0, stream>>>(src, parent, b, n);
    ccl_merge_kernel<<<grid, 256, 0, stream>>>(parent, b, n);
    ccl_compress_kernel<<<grid, 256, 0, stream>>>(parent, n);
    ccl_count_kernel<<<nchunks, 256, 0, stream>>>(parent, n, chunk);
    ccl_scan_kernel<<<1, 1024, 0, stream>>>(chunk, nchunks, state);
    ccl_rank_kernel<<<nchunks, 256, 0, stream>>>(parent, n, chunk, state);
    ccl_write_kernel<<<grid, 256, 0, stream>>>(parent, labels, b, n);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_seam_pairs(const int32_t* labels, int X, int Y, int Z, int axis, int v, int32_t* pairs,
                  int32_t* count, int capacity, void* stream) {
    SK_CHECK_ARG(labels && pairs && count && capacity > 0, "sk_seam_pairs: bad arguments");
    SK_CHECK_ARG(axis >= 0 && axis <= 2, "sk_seam_pairs: axis must be 0..2");
    int dim = axis == 0 ? X : (axis == 1 ? Y : Z);
    SK_CHECK_ARG(v >= 1 && v < dim, "sk_seam_pairs: plane %d outside (0,%d)", v, dim);
    long long n = axis == 0 ? (long long)Y * Z : (axis == 1 ? (long long)X * Z : (long long)X * Y);
    seam_pairs_kernel<<<sk::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(labels, X, Y, Z, axis, v,
                                                                         pairs, count, capacity);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_seam_components_host(const int32_t* pairs, int n_pairs, int32_t* to_replace,
                            int32_t* replace_with, int capacity) {
    SK_CHECK_ARG(n_pairs >= 0 && (n_pairs == 0 || pairs) && to_replace && replace_with,
                 "sk_seam_components_host: bad arguments");
    // graph with insertion-ordered nodes and edge lists (flood_fill.py:82-91)
    std::unordered_map<int, int> slot;
    std::vector<int> node;
    std::vector<std::vector<int>> adj;
    auto touch = [&](int v) {
        auto it = slot.find(v);
        if (it != slot.end()) return it->second;
        int s = (int)node.size();
        slot.emplace(v, s);
        node.push_back(v);
        adj.emplace_back();
        return s;
    };
    for (int i = 0; i < n_pairs; ++i) {
        int a = pairs[2 * i], b = pairs[2 * i + 1];
        int sa = touch(a);
        adj[sa].push_back(b);
        int sb = touch(b);
        adj[sb].push_back(a);
    }
    // depth-first pre-order, nodes in insertion order, edges in list order (:143-174)
    std::vector<char> seen(node.size(), 0);
    std::vector<int> comp;
    std::vector<std::pair<int, size_t>> stack;
    int written = 0;
    for (size_t s0 = 0; s0 < node.size(); ++s0) {
        if (seen[s0]) continue;
        comp.clear();
        seen[s0] = 1;
        comp.push_back(node[s0]);
        stack.clear();
        stack.emplace_back((int)s0, 0);
        while (!stack.empty()) {
            auto [s, i] = stack.back();
            stack.pop_back();
            const std::vector<int>& nb = adj[s];
            while (i < nb.size() && seen[slot[nb[i]]]) ++i;
            if (i < nb.size()) {
                stack.emplace_back(s, i + 1);
                int t = slot[nb[i]];
                seen[t] = 1;
                comp.push_back(node[t]);
                stack.emplace_back(t, 0);
            }
        }
        int keep = comp.back();  // :101-105 the LAST id of the component represents it
        for (size_t k = 0; k + 1 < comp.size(); ++k) {
            if (written >= capacity) {
                sk::set_error("sk_seam_components_host: capacity %d too small", capacity);
                return SK_ERR_CAPACITY;
            }
            to_replace[written] = comp[k];
            replace_with[written] = keep;
            ++written;
        }
    }
    return written;
}

int sk_relabel_lut(int32_t* labels, int64_t n, const int32_t* lut, int lut_size, void* stream) {
    SK_CHECK_ARG(labels && lut && n > 0 && lut_size > 0, "sk_relabel_lut: bad arguments");
    relabel_lut_kernel<<<sk::stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(labels, n, lut,
                                                                                 lut_size);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

}  // extern "C"
