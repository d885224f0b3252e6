// Error plumbing of libgg (thread-local last-error string).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/gg.h"

static thread_local char g_err[1024] = "";

void gg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* gg_last_error(void) { return g_err; }
extern "C" int gg_version(void) { return GG_VERSION; }
