// global_load_lds_dwordx4 placement on gfx950: lane i's 16 bytes land at (wave-uniform LDS base) + 16*i.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_probe.hip -o /tmp/dma && /tmp/dma
// Each lane fetches element (lane ^ 1) of its wave's 64-element slice; with lane-linear placement the LDS
// image is the slice with neighbours swapped.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint4* src, uint4* out) {
    __shared__ __attribute__((aligned(16))) uint4 S[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef __attribute__((address_space(1))) const void* gp;
    typedef __attribute__((address_space(3))) void* lp;
    __builtin_amdgcn_global_load_lds((gp)(src + blockIdx.x * 256 + wave * 64 + (lane ^ 1)), (lp)(S + wave * 64), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = S[threadIdx.x];
}
int main() {
    const int n = 256 * 4;
    std::vector<uint4> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = make_uint4(i, i + 1000, i + 2000, i + 3000);
    uint4 *d, *e;
    hipMalloc(&d, n * 16); hipMalloc(&e, n * 16);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(4), dim3(256), 0, 0, d, e);
    hipMemcpy(o.data(), e, n * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        int want = (i & ~63) + ((i & 63) ^ 1);
        if ((int)o[i].x != want || (int)o[i].w != want + 3000) { if (bad < 5) printf("i=%d got %u want %d\n", i, o[i].x, want); ++bad; }
    }
    printf("lds dma probe: %d mismatches\n", bad);
    return bad != 0;
}
