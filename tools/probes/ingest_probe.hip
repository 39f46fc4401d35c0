// How fast can a CU pull data in, as a function of resident workgroups and loads in flight per thread?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/ingest_probe.hip -o /tmp/ingest && /tmp/ingest
// Each 256-thread workgroup streams `steps` tiles of U x 16 B per thread (coalesced 4 KB rows per wave
// instruction), either from its own 64 KB window of a small buffer (L2-resident after the first touch) or
// from a private slice of a 2 GB buffer (HBM).  STAGE = 1 mimics the GEMM loaders: registers -> LDS ->
// __syncthreads() each step (one tile in flight per workgroup); STAGE = 0 just accumulates (the U loads
// of a step are independent, the next step's loads can issue immediately).
// Occupancy is set with dynamic LDS (160 KB / lds_bytes workgroups per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int U, int STAGE>
__global__ __launch_bounds__(256) void ingest(const uint4* __restrict__ src, long wg_stride, long wrap, int steps,
                                              uint4* __restrict__ sink) {
    extern __shared__ uint4 lds[];
    const uint4* base = src + (long)blockIdx.x * wg_stride;
    uint4 acc = make_uint4(0, 0, 0, 0);
    long off = threadIdx.x;
    for (int s = 0; s < steps; ++s) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = base[(off + u * 256) % wrap];
        off += U * 256;
        if (STAGE) {
#pragma unroll
            for (int u = 0; u < U; ++u) lds[threadIdx.x + u * 256] = v[u];
            __syncthreads();
            uint4 w = lds[(threadIdx.x * 7 + s) & 255];
            acc.x ^= w.x;
            __syncthreads();
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
    }
    if (acc.x == 0x12345678u) sink[blockIdx.x] = acc;     // never true in practice: keeps the loads alive
}

template <int U, int STAGE>
static double run(const uint4* src, long wg_stride, long wrap, int steps, int grid, int lds_bytes, uint4* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)ingest<U, STAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((ingest<U, STAGE>), dim3(grid), dim3(256), lds_bytes, 0, src, wg_stride, wrap, steps, sink);
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((ingest<U, STAGE>), dim3(grid), dim3(256), lds_bytes, 0, src, wg_stride, wrap, steps, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)grid * steps * U * 256 * 16 * reps;
    return bytes / (ms * 1e-3) / 1e12;
}

int main() {
    const long big = 2L << 30;                    // 2 GB
    uint4 *buf, *sink;
    hipMalloc(&buf, big); hipMalloc(&sink, 1 << 20);
    hipMemset(buf, 1, big);
    printf("%-6s %-5s %-3s %-9s %10s %14s\n", "source", "stage", "U", "WGs/CU", "TB/s", "B/clk/CU@2.4G");
    for (int hbm = 0; hbm < 2; ++hbm)
        for (int stage = 0; stage < 2; ++stage)
            for (int wpc = 1; wpc <= 8; wpc *= 2) {
                const int lds_bytes = 160 * 1024 / wpc - (wpc > 1 ? 1024 : 0);
                const int grid = 256 * wpc * 4;   // 4 rounds
                for (int U : {1, 2, 4, 8}) {
                    // L2-resident: every workgroup re-reads its own 64 KB window (256 CUs * 8 * 64 KB = 128 MB > L2,
                    // so windows are shared by workgroup index mod 64: 4 MB total, fits one XCD's L2)
                    long wrap = hbm ? (long)(big / 16 / grid) : 4096;          // in uint4
                    long stride = hbm ? wrap : 4096L * (0);                    // L2 mode: all workgroups, one window set
                    int steps = hbm ? (int)(wrap / (U * 256)) : 64 * 8 / U;
                    if (hbm && steps > 256) steps = 256;
                    double t = 0;
                    const uint4* src = hbm ? buf : buf;
                    if (!hbm) stride = 0;
                    if (U == 1) t = stage ? run<1, 1>(src, stride, wrap, steps, grid, lds_bytes, sink) : run<1, 0>(src, stride, wrap, steps, grid, lds_bytes, sink);
                    if (U == 2) t = stage ? run<2, 1>(src, stride, wrap, steps, grid, lds_bytes, sink) : run<2, 0>(src, stride, wrap, steps, grid, lds_bytes, sink);
                    if (U == 4) t = stage ? run<4, 1>(src, stride, wrap, steps, grid, lds_bytes, sink) : run<4, 0>(src, stride, wrap, steps, grid, lds_bytes, sink);
                    if (U == 8) t = stage ? run<8, 1>(src, stride, wrap, steps, grid, lds_bytes, sink) : run<8, 0>(src, stride, wrap, steps, grid, lds_bytes, sink);
                    printf("%-6s %-5d %-3d %-9d %10.2f %14.1f\n", hbm ? "HBM" : "L2", stage, U, wpc, t, t * 1e12 / 256 / 2.4e9);
                }
            }
    return 0;
}
