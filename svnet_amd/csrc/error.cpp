// Thread-local error text behind svnet_last_error(); nothing throws across the C ABI.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/svnet_hip.h"

static thread_local char g_err[512] = "";

void svnet_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* svnet_last_error(void) { return g_err; }
extern "C" int svnet_version(void) { return SVNET_ABI_VERSION; }
