// Renumber instance labels to 1..K in order of first appearance (C order), 0 kept.
//
// Replaces the fastremap.renumber(instance_mask, in_place=True) call at
// skoots/lib/eval.py:304-306 (third-party Cython, absent from the reference tree;
// semantics restated in oracle/pipeline.py:renumber, "parity unpinned").
//
// HBM-bound: one read pass (first occurrence per label via atomicMin on run
// starts), a bitmap of first-occurrence positions ranked with a popcount prefix
// sum, and one read+write relabel pass.
#include "common.h"

namespace {

constexpr int kChunkWords = 2048;
constexpr unsigned kNone = 0xFFFFFFFFu;

__global__ void __launch_bounds__(256) first_seen_kernel(const int32_t* __restrict__ labels,
                                                         long long n, int max_label,
                                                         unsigned* __restrict__ first) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int v = labels[i];
        if (v <= 0 || v > max_label) continue;
        if (i > 0 && labels[i - 1] == v) continue;  // only run starts can be a first occurrence
        if (first[v] > (unsigned)i) atomicMin(&first[v], (unsigned)i);
    }
}

__global__ void __launch_bounds__(256) first_seen_slab_kernel(const int32_t* __restrict__ labels,
                                                              int Y, int zl, int z_off, int Zg,
                                                              long long n, int max_label,
                                                              unsigned* __restrict__ first) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int v = labels[i];
        if (v <= 0 || v > max_label) continue;
        int z = (int)(i % zl);
        if (z > 0 && labels[i - 1] == v) continue;
        long long xy = i / zl;
        unsigned gidx = (unsigned)(xy * Zg + z_off + z);
        if (first[v] > gidx) atomicMin(&first[v], gidx);
    }
}

__global__ void __launch_bounds__(256) mark_kernel(const unsigned* __restrict__ first, int max_label,
                                                   unsigned* __restrict__ bitmap) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < 1 || v > max_label) return;
    unsigned f = first[v];
    if (f != kNone) atomicOr(&bitmap[f >> 5], 1u << (f & 31));
}

__global__ void __launch_bounds__(256) chunk_popc_kernel(const unsigned* __restrict__ bitmap,
                                                         long long nwords,
                                                         int* __restrict__ chunk_sum) {
    __shared__ int wsum[4];
    long long base = (long long)blockIdx.x * kChunkWords;
    int c = 0;
    for (int k = threadIdx.x; k < kChunkWords; k += 256) {
        long long w = base + k;
        if (w < nwords) c += __popc(bitmap[w]);
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void __launch_bounds__(1024) chunk_scan_kernel(int* __restrict__ chunk_sum, int nchunks,
                                                          int32_t* __restrict__ total) {
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nchunks; base += 1024) {
        int i = base + threadIdx.x;
        int v = (i < nchunks) ? chunk_sum[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            int t = (threadIdx.x >= o) ? part[threadIdx.x - o] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        int incl = part[threadIdx.x];
        if (i < nchunks) chunk_sum[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

// exclusive popcount prefix per bitmap word
__global__ void __launch_bounds__(256) word_prefix_kernel(const unsigned* __restrict__ bitmap,
                                                          long long nwords,
                                                          const int* __restrict__ chunk_off,
                                                          int* __restrict__ prefix) {
    __shared__ int wpre[4];
    __shared__ int run;
    long long base = (long long)blockIdx.x * kChunkWords;
    if (threadIdx.x == 0) run = chunk_off[blockIdx.x];
    __syncthreads();
    for (int k = 0; k < kChunkWords; k += 256) {
        long long w = base + k + threadIdx.x;
        int c = (w < nwords) ? __popc(bitmap[w]) : 0;
        int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        int incl = c;
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wpre[wv] = incl;
        __syncthreads();
        int before = 0;
        for (int q = 0; q < wv; ++q) before += wpre[q];
        int tot = wpre[0] + wpre[1] + wpre[2] + wpre[3];
        int r0 = run;
        if (w < nwords) prefix[w] = r0 + before + incl - c;
        __syncthreads();
        if (threadIdx.x == 0) run = r0 + tot;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) build_lut_kernel(const unsigned* __restrict__ first,
                                                        int max_label,
                                                        const unsigned* __restrict__ bitmap,
                                                        const int* __restrict__ prefix,
                                                        int32_t* __restrict__ lut) {
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > max_label) return;
    int r = 0;
    if (v >= 1) {
        unsigned f = first[v];
        if (f != kNone) {
            unsigned word = bitmap[f >> 5];
            r = prefix[f >> 5] + __popc(word & ((1u << (f & 31)) - 1u)) + 1;
        }
    }
    lut[v] = r;
}

__global__ void __launch_bounds__(256) apply_lut_kernel(int32_t* __restrict__ labels, long long n,
                                                        const int32_t* __restrict__ lut,
                                                        int max_label) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int v = labels[i];
        if (v > 0 && v <= max_label) labels[i] = lut[v];
    }
}

struct Layout {
    size_t off_first, off_lut, off_bitmap, off_prefix, off_chunk, total;
    long long nwords;
    int nchunks;
};

Layout layout(int64_t n, int max_label) {
    Layout L;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    L.nwords = (n + 31) / 32;
    L.nchunks = (int)((L.nwords + kChunkWords - 1) / kChunkWords);
    size_t tbl = al(((size_t)max_label + 1) * 4);
    L.off_first = 0;
    L.off_lut = L.off_first + tbl;
    L.off_bitmap = L.off_lut + tbl;
    L.off_prefix = L.off_bitmap + al((size_t)L.nwords * 4);
    L.off_chunk = L.off_prefix + al((size_t)L.nwords * 4);
    L.total = L.off_chunk + al(((size_t)L.nchunks + 16) * 4);
    return L;
}

}  // namespace

extern "C" {

size_t sk_renumber_workspace_bytes(int64_t n, int max_label) {
    if (n <= 0 || max_label < 0) return 0;
    return layout(n, max_label).total;
}

int sk_first_seen(const int32_t* labels, int X, int Y, int zl, int z_off, int Zg, int max_label,
                  uint32_t* first, void* stream) {
    SK_CHECK_ARG(labels && first, "sk_first_seen: NULL pointer");
    SK_CHECK_ARG(X > 0 && Y > 0 && zl > 0 && z_off >= 0 && z_off + zl <= Zg && max_label >= 0,
                 "sk_first_seen: bad extents");
    SK_CHECK_ARG((long long)X * Y * Zg <= 0xFFFFFFFELL, "sk_first_seen: volume too large");
    long long n = (long long)X * Y * zl;
    first_seen_slab_kernel<<<sk::stream_grid(n, 256), 256, 0, (hipStream_t)stream>>>(
        labels, Y, zl, z_off, Zg, n, max_label, first);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

int sk_renumber(int32_t* labels, int64_t n, int max_label, void* workspace, size_t workspace_bytes,
                int32_t* n_labels, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SK_CHECK_ARG(labels && workspace && n_labels, "sk_renumber: NULL pointer");
    SK_CHECK_ARG(n > 0 && n <= 0xFFFFFFFELL, "sk_renumber: n=%lld out of range", (long long)n);
    SK_CHECK_ARG(max_label >= 0, "sk_renumber: max_label must be >= 0");
    Layout L = layout(n, max_label);
    SK_CHECK_ARG(workspace_bytes >= L.total, "sk_renumber: workspace too small (%zu < %zu)",
                 workspace_bytes, L.total);
    char* ws = (char*)workspace;
    unsigned* first = (unsigned*)(ws + L.off_first);
    int32_t* lut = (int32_t*)(ws + L.off_lut);
    unsigned* bitmap = (unsigned*)(ws + L.off_bitmap);
    int* prefix = (int*)(ws + L.off_prefix);
    int* chunk = (int*)(ws + L.off_chunk);
    SK_CHECK_HIP(hipMemsetAsync(first, 0xFF, ((size_t)max_label + 1) * 4, stream));
    SK_CHECK_HIP(hipMemsetAsync(bitmap, 0, (size_t)L.nwords * 4, stream));
    first_seen_kernel<<<sk::stream_grid(n, 256), 256, 0, stream>>>(labels, n, max_label, first);
    unsigned tgrid = sk::cdiv((long long)max_label + 1, 256);
    mark_kernel<<<tgrid, 256, 0, stream>>>(first, max_label, bitmap);
    chunk_popc_kernel<<<L.nchunks, 256, 0, stream>>>(bitmap, L.nwords, chunk);
    chunk_scan_kernel<<<1, 1024, 0, stream>>>(chunk, L.nchunks, n_labels);
    word_prefix_kernel<<<L.nchunks, 256, 0, stream>>>(bitmap, L.nwords, chunk, prefix);
    build_lut_kernel<<<tgrid, 256, 0, stream>>>(first, max_label, bitmap, prefix, lut);
    apply_lut_kernel<<<sk::stream_grid(n, 256), 256, 0, stream>>>(labels, n, lut, max_label);
    SK_CHECK_LAUNCH();
    return SK_OK;
}

}  // extern "C"
