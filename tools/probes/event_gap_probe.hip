// What does a cross-stream hand-off cost the PRODUCING stream?  ttsmi_dense_block_bwd records 4 events per block on the
// main stream (hipEventRecord + hipStreamWaitEvent on the weight-gradient stream); the kernel trace shows ~6 us of idle
// main-stream time after every one of them (52 per step = 0.33 ms), while kernels without a record in between start
// back to back.  Modes:
//   0  N kernels back to back on stream A (no events)
//   1  hipEventRecord(e, A) after every kernel, nobody waits
//   2  hipEventRecord(e, A) + hipStreamWaitEvent(B, e) + a small kernel on B   (what the library does today)
//   3  hipExtLaunchKernelGGL(..., stopEvent = e) - the event rides on the kernel's own completion signal - + wait + B kernel
//   4  as 2 but one record for every 2nd kernel
// Prints us per kernel of stream A (HIP events around the whole sequence) and checks that B really ran after A's kernel.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/event_gap_probe.hip -o tools/probes/event_gap_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>

__global__ void busy(float* p, int iters, int tag) {
    float v = p[threadIdx.x] + tag;
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) p[0] = v;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) p[1024 + tag % 1024] = (float)tag;
}
__global__ void consume(const float* p, float* out, int tag) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[tag % 1024] = p[1024 + tag % 1024];      // must see the producer's tag
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int N = 400, ITERS = 1500;
    float *p, *out;
    CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
    CK(hipMalloc(&out, 1024 * sizeof(float)));
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int mode = 0; mode <= 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            const int base = 1024 * (mode * 2 + rep + 1);          // a fresh tag range: a consumer that ran early reads an old tag
            CK(hipMemsetAsync(out, 0, N * sizeof(float), A));
            CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
            CK(hipEventRecord(t0, A));
            for (int i = 0; i < N; ++i) {
                if (mode == 3) {
                    hipExtLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, A, nullptr, ev[i], 0, p, ITERS, i + base);
                } else {
                    hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, A, p, ITERS, i + base);
                    if (mode == 1 || mode == 2 || (mode == 4 && (i & 1))) CK(hipEventRecord(ev[i], A));
                }
                if (mode == 2 || mode == 3 || (mode == 4 && (i & 1))) {
                    CK(hipStreamWaitEvent(B, ev[i], 0));
                    hipLaunchKernelGGL(consume, dim3(1), dim3(64), 0, B, p, out + 0, i + base);
                }
            }
            CK(hipEventRecord(t1, A));
            CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, t0, t1));
            int bad = 0;
            if (mode >= 2) {
                std::vector<float> h(N);
                CK(hipMemcpy(h.data(), out, N * sizeof(float), hipMemcpyDeviceToHost));
                for (int i = 0; i < N; ++i) if ((mode != 4 || (i & 1)) && h[i] != (float)(i + base)) ++bad;
            }
            if (rep) printf("mode %d: %.2f us per kernel on stream A, consumer mismatches %d\n", mode, ms * 1e3 / N, bad);
        }
    }
    return 0;
}
