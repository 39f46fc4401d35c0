// Do the matrix pipe and the vector pipe of a SIMD overlap ACROSS waves?  The attention forward's stage ablation is
// additive (removing the softmax algebra gives back its VALU issue time, removing a product its MFMA pipe time), as if
// nothing overlapped although 4 waves share a SIMD.  Each wave runs  loop { NM dependent-free MFMAs ; NV VALU fmas }:
//   mode 0: MFMA only     mode 1: VALU only     mode 2: both, MFMA block then VALU block (phases, as the kernels do)
//   mode 3: both, interleaved in the instruction stream (1 MFMA : NV/NM VALU)
// run with 1, 2 and 4 waves per SIMD (256-thread workgroups, 1 / 2 / 4 per CU via dynamic LDS).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_valu_overlap_probe.hip -o tools/probes/mfma_valu_overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    extern __shared__ char pad[];
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = threadIdx.x * 0.001f + r;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.01f + e); b[e] = (__bf16)(e * 0.5f); }
    float v[16];
    for (int r = 0; r < 16; ++r) v[r] = threadIdx.x + r * 0.25f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = __builtin_fmaf(v[r], 1.0001f, 0.5f);
        }
        if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 12; ++r) v[(j * 12 + r) & 15] = __builtin_fmaf(v[(j * 12 + r) & 15], 1.0001f, 0.5f);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int r = 0; r < 16; ++r) s += v[r];
    if (s == 12345.678f) out[0] = s;
    (void)pad;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int MODE>
static float run(int wgs_per_cu, float* out) {
    const int iters = 2000;
    const size_t lds = wgs_per_cu == 4 ? 32 * 1024 : wgs_per_cu == 2 ? 64 * 1024 : 128 * 1024;   // caps the workgroups per CU
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    probe<MODE><<<256 * wgs_per_cu, 256, lds>>>(out, 10);
    hipEventRecord(t0);
    probe<MODE><<<256 * wgs_per_cu, 256, lds>>>(out, iters);
    hipEventRecord(t1);
    hipEventSynchronize(t1);
    float ms = 0;
    hipEventElapsedTime(&ms, t0, t1);
    return ms * 1e6f / iters;          // ns per loop iteration (all waves in parallel)
}
int main() {
    float* out;
    CK(hipMalloc(&out, 64));
    printf("per iteration: 8 MFMA 32x32x16 bf16 (256 matrix-pipe cycles) and 96 VALU fma (384 issue cycles) per wave\n");
    for (int w : {1, 2, 4}) {
        printf("%d wave(s)/SIMD: mfma only %.0f ns, valu only %.0f ns, phases %.0f ns, interleaved %.0f ns\n", w,
               run<0>(w, out), run<1>(w, out), run<2>(w, out), run<3>(w, out));
    }
    return 0;
}
