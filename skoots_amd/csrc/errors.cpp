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
}  // namespace sk

extern "C" {
const char* sk_last_error(void) { return sk::g_err; }
int sk_abi_version(void) { return 2; }  // 2: round 2 (split mode, fused down conv, bf16 twins; sk_conv3d ksize 2 needs the zero page)
}
