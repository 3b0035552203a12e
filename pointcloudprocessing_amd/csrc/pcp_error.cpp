#include <stdarg.h>
#include <stdio.h>
#include <string.h>

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

void pcp_append_error(const char* text) {
    const size_t have = strlen(g_err);
    snprintf(g_err + have, sizeof(g_err) - have, "%s", text);
}

extern "C" const char* pcp_last_error(void) { return g_err; }
