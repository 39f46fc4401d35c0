// Issue-rate probe for the VALU instructions that sit in the attention inner loop (gfx950): N dependent-free copies of
// one instruction per loop iteration on 8 independent registers, one wave per SIMD vs 2 / 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe tools/probes/valu_rate_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void probe(uint32_t* out, int iters, uint32_t seed) {
    uint32_t r0 = seed + threadIdx.x, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3, r4 = r0 ^ 0x55, r5 = r0 ^ 0xaa, r6 = r0 + 77, r7 = r0 + 99;
    uint32_t c = seed | 1;
    for (int i = 0; i < iters; ++i) {
#define ASM1(OPS, R) asm volatile(OPS : "+v"(R) : "v"(c));
        if (OP == 0) { // v_mul_lo_u32
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r##i) : "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 1) { // v_xor_b32
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r##i) : "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 2) { // v_mul_u32_u24
#define X(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r##i) : "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 3) { // v_exp_f32
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r##i));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 4) { // v_mad_u32_u24
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(r##i) : "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 5) { // v_cndmask
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##i) : "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 6) { // v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r##i) : "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 7) { // v_mul_hi_u32
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r##i) : "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 8) { // v_lshl_add_u32 / xad
#define X(i) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(r##i) : "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 9) { // v_alignbit (rotate)
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(r##i));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 10) { // v_cmp + nothing
#define X(i) asm volatile("v_cmp_le_u32 vcc, %0, %1" :: "v"(r##i), "v"(c) : "vcc");
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 11) { // v_pk_mul_f32
            uint64_t q0 = r0, q1 = r1, q2 = r2, q3 = r3; uint64_t cc = c;
#define Y(q) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q) : "v"(cc));
            Y(q0) Y(q1) Y(q2) Y(q3) Y(q0) Y(q1) Y(q2) Y(q3) Y(q0) Y(q1) Y(q2) Y(q3) Y(q0) Y(q1) Y(q2) Y(q3)
            Y(q0) Y(q1) Y(q2) Y(q3) Y(q0) Y(q1) Y(q2) Y(q3) Y(q0) Y(q1) Y(q2) Y(q3) Y(q0) Y(q1) Y(q2) Y(q3)
#undef Y
            r0 ^= (uint32_t)q0; r1 ^= (uint32_t)q1; r2 ^= (uint32_t)q2; r3 ^= (uint32_t)q3;
        } else if (OP == 12) { // v_bfe_u32
#define X(i) asm volatile("v_bfe_u32 %0, %0, 3, 16" : "+v"(r##i));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 14) { // v_cndmask e64 with an SGPR pair mask
            uint64_t m = 0x5555aaaa5555aaaaull ^ seed;
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r##i) : "v"(c), "s"(m));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 15) { // v_cmp + v_cndmask pairs (the dropout select as compiled today)
#define X(i) asm volatile("v_cmp_le_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##i) : "v"(c) : "vcc");
            REP8(X) REP8(X)
#undef X
        } else if (OP == 16) { // v_cndmask with a destination that is not a source
            uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
#define X(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(t##i) : "v"(r##i), "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
            r0 ^= t0; r1 ^= t1; r2 ^= t2; r3 ^= t3; r4 ^= t4; r5 ^= t5; r6 ^= t6; r7 ^= t7;
        } else if (OP == 17) { // v_and_b32
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r##i) : "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 18) { // v_bfe_i32 (1-bit sign-extended field -> 0 / -1 mask)
#define X(i) asm volatile("v_bfe_i32 %0, %0, 5, 1" : "+v"(r##i));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 19) { // v_max3_f32
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(r##i) : "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 20) { // v_mul_f32 e32
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r##i) : "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 21) { // v_add_u32 e32
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r##i) : "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (OP == 13) { // v_cvt_pk_bf16_f32
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r##i) : "v"(c));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}

template <int OP>
static void run(const char* name, uint32_t* d) {
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD
        dim3 grid(256), block(256 * wps);            // 1 block per CU: 4*wps waves per CU
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<OP>, grid, block, 0, 0, d, 10, 1u);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<OP>, grid, block, 0, 0, d, iters, 1u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // cycles per wave-instruction per SIMD at 2.4 GHz: each SIMD issued wps * iters * 32 instructions
        double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * iters * 32);
        printf("%-18s waves/SIMD %d: %.2f cycles per wave-instruction (SIMD-level)\n", name, wps, cyc);
    }
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 1024 * 4);
    run<1>("v_xor_b32", d); run<0>("v_mul_lo_u32", d); run<7>("v_mul_hi_u32", d); run<2>("v_mul_u32_u24", d);
    run<4>("v_mad_u32_u24", d); run<3>("v_exp_f32", d); run<5>("v_cndmask_b32", d); run<6>("v_fma_f32", d);
    run<8>("v_xad_u32", d); run<9>("v_alignbit_b32", d); run<10>("v_cmp_le_u32", d); run<11>("v_pk_mul_f32", d);
    run<12>("v_bfe_u32", d); run<13>("v_cvt_pk_bf16_f32", d);
    run<14>("v_cndmask e64 sgpr", d); run<15>("v_cmp+v_cndmask (x16)", d); run<16>("v_cndmask dst!=src", d);
    run<17>("v_and_b32", d); run<18>("v_bfe_i32", d); run<19>("v_max3_f32", d); run<20>("v_mul_f32", d); run<21>("v_add_u32", d);
    return 0;
}
