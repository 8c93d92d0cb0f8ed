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
// the sha256 (16 hex digits) over every source this library was compiled from, set by pgtformer_amd/build.py: measurement files
// quote the stamp of the binary that ran
#ifndef PGT_SOURCE_SHA16
#define PGT_SOURCE_SHA16 "unstamped-build!"
#endif
extern "C" const char* pgt_version(void) { return "pgt_hip 0.3 (gfx950) src:" PGT_SOURCE_SHA16; }
