// libttsmi: version + thread-local error message + thread-local "which kernel variant did the last router pick".
#include <stdarg.h>
#include <stdio.h>

#include "../../include/ttsmi.h"

static thread_local char g_err[512] = "";
static thread_local const char* g_kernel = "";

// launch routers record which kernel variant they chose (string literals only): ttsmi_last_kernel()
void ttsmi_note_kernel(const char* name) { g_kernel = name; }

void ttsmi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
int ttsmi_version(void) { return TTSMI_VERSION; }
const char* ttsmi_last_error(void) { return g_err; }
const char* ttsmi_last_kernel(void) { return g_kernel; }
}
