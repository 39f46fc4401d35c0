// wave_sum_dpp (common.h) against the __shfl_xor butterfly, full waves and after a divergent region.
//   hipcc --offload-arch=gfx950 -O3 -I transformertts_amd/csrc -I include tools/probes/dpp_reduce_probe.hip -o /tmp/dpp && /tmp/dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include "common.h"
__device__ __forceinline__ float shfl_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ void k(const float* x, float* out, int C) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float s = 0.f, t = 0.f;
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j;
        if (c < C) { const float v = x[w * 256 + c]; s += v; t += v * v; }
    }
    const float a = wave_sum_dpp(s), b = shfl_sum(s), a2 = wave_sum_dpp(t), b2 = shfl_sum(t);
    const float m = wave_max(s);
    float mm = s;
    for (int o = 32; o > 0; o >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o, 64));
    out[(w * 64 + lane) * 6 + 0] = a; out[(w * 64 + lane) * 6 + 1] = b;
    out[(w * 64 + lane) * 6 + 2] = a2; out[(w * 64 + lane) * 6 + 3] = b2;
    out[(w * 64 + lane) * 6 + 4] = m; out[(w * 64 + lane) * 6 + 5] = mm;
}
int main() {
    float hx[1024], *dx, *dout, ho[4 * 64 * 6];
    for (int i = 0; i < 1024; ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMalloc(&dx, sizeof(hx)); hipMalloc(&dout, sizeof(ho));
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    for (int C : {256, 226, 80, 1}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dx, dout, C);
        hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
        double worst = 0; int bad = 0;
        for (int i = 0; i < 256; ++i) {
            for (int q = 0; q < 3; ++q) {
                const double d = fabs((double)ho[i * 6 + 2 * q] - ho[i * 6 + 2 * q + 1]) / (fabs((double)ho[i * 6 + 2 * q + 1]) + 1e-6);
                if (d > worst) worst = d;
                if (d > 1e-5) ++bad;
            }
        }
        printf("C=%d: worst relative difference dpp vs shuffle %.3e, lanes off by > 1e-5: %d  (wave 0: sum %g / %g, max %g / %g)\n", C, worst, bad,
               ho[0], ho[1], ho[4], ho[5]);
    }
    return 0;
}
