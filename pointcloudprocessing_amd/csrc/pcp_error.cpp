#include <stdarg.h>
#include <stdio.h>

#include "../../include/pcp.h"
#include "pcp_host.h"

static thread_local char g_err[1024] = "";

int pcp_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

extern "C" const char* pcp_last_error(void) { return g_err; }
