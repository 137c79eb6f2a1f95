// Version / error-string entry points of the C ABI (include/slam_amd.h).
#include "common.h"

static thread_local char g_err[512] = "";

void slam_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *slam_version(void) { return "slam_amd 0.1 (gfx950)"; }
extern "C" const char *slam_last_error_string(void) { return g_err; }
