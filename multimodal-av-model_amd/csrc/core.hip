// Error plumbing + version of libavhip.so (no global mutable state besides the thread-local error string).
#include <cstdarg>
#include <cstdio>

#include "av_common.h"

static thread_local char g_err[512] = "";

void av_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* av_last_error(void) { return g_err; }
extern "C" int av_version(void) { return 1; }
