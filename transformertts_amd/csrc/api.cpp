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

// ---- measurement only: a CU-masked stream and a census of where its workgroups land (the
// weight-gradient side stream confined to whole XCDs against today's free-for-all) ----------------------------------------
__global__ void xcc_census_kernel(int* counts) {
    if (threadIdx.x == 0) {
        // HW_REG_XCC_ID (id 20), bits 3:0 = the XCC this wave runs on
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 15u;
        atomicAdd(counts + (xcc & 7), 1);
    }
    // stay resident for a moment so that the blocks spread over every CU the stream may use
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 20000) {}
}

extern "C" {
int ttsmi_debug_stream_create_cu_mask(const uint32_t* mask, int nwords, ttsmi_stream_t* out) {
    if (!mask || nwords <= 0 || !out) {
        ttsmi_set_error("debug_stream_create_cu_mask: bad argument");
        return TTSMI_ERR_INVALID_ARG;
    }
    hipStream_t st = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, mask);
    if (e != hipSuccess) {
        ttsmi_set_error("debug_stream_create_cu_mask: %s", hipGetErrorString(e));
        return TTSMI_ERR_LAUNCH;
    }
    *out = (ttsmi_stream_t)st;
    return TTSMI_OK;
}
int ttsmi_debug_xcc_census(int32_t* counts8, int nblocks, ttsmi_stream_t stream) {
    if (!counts8 || nblocks <= 0) {
        ttsmi_set_error("debug_xcc_census: bad argument");
        return TTSMI_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(xcc_census_kernel, dim3(nblocks), dim3(64), 0, (hipStream_t)stream, counts8);
    return hipGetLastError() == hipSuccess ? TTSMI_OK : TTSMI_ERR_LAUNCH;
}
int ttsmi_version(void) { return TTSMI_VERSION; }
const char* ttsmi_last_error(void) { return g_err; }
const char* ttsmi_last_kernel(void) { return g_kernel; }
}
