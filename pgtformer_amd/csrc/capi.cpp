// Error plumbing + version for the C-ABI (no exceptions cross the boundary, no exit()).
#include <cstdarg>
#include <cstdio>

#include "pgt_internal.h"

static thread_local char g_err[512] = "";

void pgt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* pgt_last_error(void) { return g_err; }
extern "C" const char* pgt_version(void) { return "pgt_hip 0.2 (gfx950)"; }
