// libttsmi: version + thread-local error message + thread-local "which kernel variant did the last router pick".
#include <stdarg.h>
#include <stdio.h>

#include <hip/hip_runtime.h>

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

// cross-stream hand-off riding on a kernel's completion signal (csrc/common.h: TTSMI_LAUNCH_EV)
static thread_local hipEvent_t g_stop_event = nullptr;
void ttsmi_arm_stop_event(hipEvent_t e) { g_stop_event = e; }
hipEvent_t ttsmi_take_stop_event() {
    hipEvent_t e = g_stop_event;
    g_stop_event = nullptr;
    return e;
}

extern "C" {
int ttsmi_version(void) { return TTSMI_VERSION; }
const char* ttsmi_last_error(void) { return g_err; }
const char* ttsmi_last_kernel(void) { return g_kernel; }
}
