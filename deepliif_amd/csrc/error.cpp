// error.cpp -- thread-local last-error string, version, and the load-time table of runtime switches (see include/deepliif_hip.h)
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

// ---- runtime switches: the variables below are copied out of the environment when the library is loaded; launches only read the copies
enum { DL_SW_COUNT_ = 9, DL_SW_LEN = 16 };
static const char *const g_sw_names[DL_SW_COUNT_] = {"DL_CONV_S2F", "DL_CONV_S2FX3", "DL_CONV_W4X3", "DL_PACK_TILED", "DL_NO_X3_GLDS", "DL_NO_WGRAD_C4",
                                                     "DL_NO_C4_X3", "DL_CONV_S2D", "DL_CONV_DOT"};
static char g_sw_val[DL_SW_COUNT_][DL_SW_LEN];
static bool g_sw_set[DL_SW_COUNT_];

extern "C" void dl_switches_reload(void) {
    for (int i = 0; i < DL_SW_COUNT_; ++i) {
        const char *v = getenv(g_sw_names[i]);
        g_sw_set[i] = v != nullptr;
        if (v) {
            strncpy(g_sw_val[i], v, DL_SW_LEN - 1);
            g_sw_val[i][DL_SW_LEN - 1] = 0;
        }
    }
}
__attribute__((constructor)) static void dl_switches_init(void) { dl_switches_reload(); }

const char *dl_switch(int id) { return (id >= 0 && id < DL_SW_COUNT_ && g_sw_set[id]) ? g_sw_val[id] : nullptr; }

extern "C" int dl_switch_count(void) { return DL_SW_COUNT_; }
extern "C" const char *dl_switch_name(int id) { return (id >= 0 && id < DL_SW_COUNT_) ? g_sw_names[id] : nullptr; }
extern "C" int dl_half_format(void) {
#ifdef DL_H16_FP16
    return DL_HALF_FP16;
#else
    return DL_HALF_BF16;
#endif
}
extern "C" int dl_dev_build(void) {
#ifdef DL_DEV_SWITCHES
    return 1;
#else
    return 0;
#endif
}
