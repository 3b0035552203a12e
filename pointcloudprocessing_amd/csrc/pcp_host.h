// Host-side error plumbing shared by the translation units of libpcp.so.
#pragma once
#include <hip/hip_runtime.h>

// Records a printf-style message as the calling thread's last error; returns 1.
int pcp_fail(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
// Appends to the calling thread's last error (what a caller up the stack did about it).
void pcp_append_error(const char* text);

#define PCP_CHECK(expr)                                                                      \
    do {                                                                                     \
        hipError_t pcp_e_ = (expr);                                                          \
        if (pcp_e_ != hipSuccess)                                                            \
            return pcp_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(pcp_e_), __FILE__, __LINE__); \
    } while (0)
