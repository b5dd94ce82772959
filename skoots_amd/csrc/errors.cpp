// Error channel of the C ABI: sk_last_error() returns the text of the last failure on
// the calling thread (the Python host raises ValueError / RuntimeError with it).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace sk {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static long long* g_timing = nullptr;
long long* timing_buffer() { return g_timing; }
}  // namespace sk

extern "C" {
int sk_debug_set_timing_buffer(void* device_ptr, size_t bytes) {
    const size_t need = (size_t)sk::kTimingBlocks * 4 * sk::kTimingSlots * sizeof(long long);
    if (device_ptr != nullptr && bytes < need) {
        sk::set_error("sk_debug_set_timing_buffer: %zu bytes, need %zu ([4096][4][16] int64)", bytes, need);
        return SK_ERR_ARG;
    }
    sk::g_timing = (long long*)device_ptr;
    return SK_OK;
}
const char* sk_last_error(void) { return sk::g_err; }
int sk_abi_version(void) { return 3; }  // 3: round 3 (sk_debug_set_timing_buffer, sk_mfma_probe, sk_conv3d_box, the 16-bit gradient hand-offs sk_train_interleave2_h / sumpool2_hh / heads_dgrad_f16); 2: round 2 (split mode, fused down conv, bf16 twins)
}
