// error.cpp -- thread-local last-error string + version (see include/deepliif_hip.h)
#include <stdarg.h>
#include <stdio.h>
#include "../../include/deepliif_hip.h"

static thread_local char g_err[512] = "";

void dl_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *dl_last_error(void) { return g_err; }
extern "C" int dl_version(void) { return DL_VERSION; }
