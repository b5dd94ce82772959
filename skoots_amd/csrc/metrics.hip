// Instance-level validation metrics: the N x M IoU matrix of two instance masks.
// Replaces skoots/validate/lib.py:190-229 (mask_iou: a Python double loop over instances with a full-volume
// logical_and / logical_or per touching pair) by ONE pass over the two volumes -- a contingency table of
// (ground-truth instance, predicted instance) voxel counts -- and a tiny kernel over the table.
#include "common.h"

namespace {

// table ((N+1) x (M+1)) int32, row / column 0 = background; the (0, 0) cell is not counted (never needed, and it
// would serialise the atomics of every background voxel on one address)
__global__ void __launch_bounds__(256) contingency_kernel(const int* __restrict__ a, const int* __restrict__ b, long long n,
                                                          const int* __restrict__ lut_a, int max_a,
                                                          const int* __restrict__ lut_b, int max_b, int M1,
                                                          int* __restrict__ table) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int va = a[i], vb = b[i];
        const int ra = (va > 0 && va <= max_a) ? lut_a[va] : 0;
        const int rb = (vb > 0 && vb <= max_b) ? lut_b[vb] : 0;
        if (ra | rb) atomicAdd(&table[(long long)ra * M1 + rb], 1);
    }
}

__global__ void __launch_bounds__(256) table_sums_kernel(const int* __restrict__ table, int N1, int M1,
                                                         long long* __restrict__ row, long long* __restrict__ col) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < N1) {
        long long s = 0;
        for (int j = 0; j < M1; ++j) s += table[(long long)t * M1 + j];
        row[t] = s;
    } else if (t < N1 + M1) {
        const int j = t - N1;
        long long s = 0;
        for (int i = 0; i < N1; ++i) s += table[(long long)i * M1 + j];
        col[j] = s;
    }
}

__global__ void __launch_bounds__(256) iou_kernel(const int* __restrict__ table, const long long* __restrict__ row,
                                                  const long long* __restrict__ col, int N, int M, float* __restrict__ iou) {
    const long long n = (long long)N * M;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
        const int i = (int)(t / M), j = (int)(t % M);
        const long long inter = table[(long long)(i + 1) * (M + 1) + (j + 1)];
        const long long uni = row[i + 1] + col[j + 1] - inter;
        iou[t] = inter > 0 ? (float)inter / (float)uni : 0.0f;   // lib.py:224: int / int -> fp32
    }
}

}  // namespace

extern "C" {

size_t sk_mask_iou_workspace_bytes(int N, int M) {
    return ((size_t)(N + 1) * (M + 1) * sizeof(int) + 15) / 16 * 16 + (size_t)(N + 1 + M + 1) * sizeof(long long);
}

int sk_mask_iou(const int32_t* gt, const int32_t* pred, int64_t n, const int32_t* lut_gt, int max_gt, int N,
                const int32_t* lut_pred, int max_pred, int M, float* iou, void* workspace, size_t workspace_bytes,
                void* stream) {
    SK_CHECK_ARG(gt && pred && lut_gt && lut_pred && workspace && n >= 1 && N >= 0 && M >= 0, "sk_mask_iou: bad arguments");
    SK_CHECK_ARG(workspace_bytes >= sk_mask_iou_workspace_bytes(N, M), "sk_mask_iou: workspace too small");
    SK_CHECK_ARG(N == 0 || M == 0 || iou, "sk_mask_iou: NULL iou");
    hipStream_t st = (hipStream_t)stream;
    int* table = (int*)workspace;
    const size_t tbytes = ((size_t)(N + 1) * (M + 1) * sizeof(int) + 15) / 16 * 16;
    long long* row = (long long*)((char*)workspace + tbytes);
    long long* col = row + (N + 1);
    SK_CHECK_HIP(hipMemsetAsync(table, 0, tbytes, st));
    contingency_kernel<<<sk::stream_grid(n, 256, 4), 256, 0, st>>>(gt, pred, n, lut_gt, max_gt, lut_pred, max_pred, M + 1, table);
    SK_CHECK_LAUNCH();
    table_sums_kernel<<<sk::cdiv(N + M + 2, 256), 256, 0, st>>>(table, N + 1, M + 1, row, col);
    SK_CHECK_LAUNCH();
    if (N > 0 && M > 0) {
        iou_kernel<<<sk::stream_grid((long long)N * M, 256, 1), 256, 0, st>>>(table, row, col, N, M, iou);
        SK_CHECK_LAUNCH();
    }
    return SK_OK;
}

}  // extern "C"
