// libttsmi: version + thread-local error message.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/ttsmi.h"

static thread_local char g_err[512] = "";

void ttsmi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
int ttsmi_version(void) { return TTSMI_VERSION; }
const char* ttsmi_last_error(void) { return g_err; }
}
