// Error plumbing shared by every C-ABI entry point (see include/gd_hip.h).
#include "gd_common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void gd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gd_last_error(void) { return g_err; }
extern "C" int gd_abi_version(void) { return 1; }
