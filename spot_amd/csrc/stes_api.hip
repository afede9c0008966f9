// Version and error reporting shared by every entry point of libstes_hip.
#include "stes_common.h"
#include <string.h>

static thread_local char g_err[256] = "";

int stes_set_error(int code, const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
    return code;
}

extern "C" int stes_version(void) { return 1; }
extern "C" const char* stes_last_error(void) { return g_err; }
