// (Round 4: the measurement that made the one-pass attention backward a TICKET scheme instead of this chain - every
// hand-off here costs a poll round trip, a tile round trip and a store acknowledgement, and start order inside an XCD is
// not block-id order; what the kernel kept from it: XCC_ID == blockIdx % 8, and plain stores + L1-bypassing loads are
// coherent inside one XCD's L2 while the same protocol across XCDs reads stale data.)
// Can workgroups of ONE kernel hand a tile to each other through L2 in a FIXED order, cheaply?  (The one-pass attention
// backward needs it: key-stationary workgroups each hold a partial dQ tile that has to be summed over the key tiles of a
// head in a reproducible order - csrc/attention_bf16.hip, hattn_bwd_fused_kernel.)
//
//  part 1 - placement: XCC_ID (s_getreg) of every workgroup against blockIdx % 8, and the order in which workgroups
//           start (a global ticket) against blockIdx - the scheme waits only for LOWER block ids of the same XCD.
//  part 2 - chain: groups of NKT workgroups (same blockIdx % 8 unless `spread`) pass NQT tiles of 16 KB down the chain
//           j = 0 .. NKT-1: wait for flag == j (one lane spins on a relaxed agent-scope load), load the tile, add,
//           store, s_waitcnt vmcnt(0), barrier, flag = j + 1 (relaxed agent-scope store).  The LAST workgroup checks the
//           exact integer sum.  Load / store cache-policy variants:
//             ld 0 plain (may hit a stale line in the CU's vector L1)   ld 1 `sc1`   ld 2 `sc0 sc1`
//             st 0 plain (write-through L1 -> L2)                        st 1 `sc0 sc1`
//           `work` = busy iterations per tile (stands in for the score arithmetic), `chain` 0 = no hand-off at all
//           (every workgroup writes its own tile: the cost of the chain is the difference).
//           Spins are BOUNDED: a wait that never ends raises the timeout count instead of hanging the GPU.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/xcd_sem_probe.hip -o tools/probes/xcd_sem_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void placement(int* xcc, int* ticket_of, int* ticket) {
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc[blockIdx.x] = (int)(x & 15);
        ticket_of[blockIdx.x] = atomicAdd(ticket, 1);
    }
    // a little work so that later blocks are dispatched while earlier ones still run
    float v = threadIdx.x;
    for (int i = 0; i < 2000; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) xcc[0] = -1;
}

struct ChainP {
    float* acc;        // [groups][nqt][4096]
    int* sem;          // [groups][nqt]
    int* err;          // [0] wrong sums, [1] spin timeouts, [2] xcc mismatches inside a group
    int* xcc_of;       // [groups] xcc id of the group's first workgroup
    int groups, nkt, nqt, work, ld, st, chain, spread;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld_tile(const float* p, int mode) {
    f32x4 v;
    if (mode == 0) v = *reinterpret_cast<const f32x4*>(p);
    else if (mode == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_tile(float* p, f32x4 v, int mode) {
    if (mode == 0) *reinterpret_cast<f32x4*>(p) = v;
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

__global__ __launch_bounds__(256) void chain(ChainP p) {
    const int bid = blockIdx.x, tid = threadIdx.x;
    int g, j;
    if (p.spread) { g = bid / p.nkt; j = bid % p.nkt; }                       // neighbours: eight different XCDs
    else { const int x = bid & 7, slot = bid >> 3; g = x + 8 * (slot / p.nkt); j = slot % p.nkt; }
    if (g >= p.groups) return;
    unsigned xid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xid));
    xid &= 15;
    if (tid == 0) {
        if (j == 0) __hip_atomic_store(p.xcc_of + g, (int)xid + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float mine = 0.f;
    for (int i = 0; i < p.nqt; ++i) {
        float v = tid * 0.001f;
        for (int k = 0; k < p.work; ++k) v = v * 1.0001f + 0.5f;
        mine = (v == 12345.f) ? 1.f : 0.f;                                     // 0: keeps the loop alive
        float* tile = p.acc + ((long)(p.chain ? g : bid) * p.nqt + i) * 4096 + tid * 4;
        int* flag = p.sem + (long)g * p.nqt + i;
        if (p.chain && j > 0) {
            if (tid == 0) {
                int spins = 0;
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != j) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1 << 17)) { atomicAdd(p.err + 1, 1); break; }
                }
            }
            __syncthreads();
        }
        f32x4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p.chain && j > 0) a[u] = ld_tile(tile + u * 1024, p.ld);
            else a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float add = (float)((j + 1) * (i + 1) + (tid % 7) + u) + mine;
            a[u].x += add; a[u].y += add + 1.f; a[u].z += add + 2.f; a[u].w += add + 3.f;
        }
        const bool last = j == p.nkt - 1;
        if (p.chain && last) {
            // expected: sum_j (j+1)(i+1) + nkt * (tid % 7 + u) (+ e per component)
            int bad = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float want = (float)((i + 1) * p.nkt * (p.nkt + 1) / 2 + p.nkt * ((tid % 7) + u));
                bad += (a[u].x != want) + (a[u].y != want + p.nkt) + (a[u].z != want + 2.f * p.nkt) + (a[u].w != want + 3.f * p.nkt);
            }
            if (bad) atomicAdd(p.err + 0, bad);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) st_tile(tile + u * 1024, a[u], p.st);
        if (p.chain) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (j > 0) {
                    const int first = __hip_atomic_load(p.xcc_of + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (first != (int)xid + 1 && i == 0) atomicAdd(p.err + 2, 1);
                }
                __hip_atomic_store(flag, last ? 0 : j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

int main(int argc, char** argv) {
    // ---- part 1
    const int NB = 4096;
    int *xcc, *tof, *ticket;
    CK(hipMalloc(&xcc, NB * 4)); CK(hipMalloc(&tof, NB * 4)); CK(hipMalloc(&ticket, 4));
    CK(hipMemset(ticket, 0, 4));
    hipLaunchKernelGGL(placement, dim3(NB), dim3(256), 0, 0, xcc, tof, ticket);
    CK(hipDeviceSynchronize());
    std::vector<int> hx(NB), ht(NB);
    CK(hipMemcpy(hx.data(), xcc, NB * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ht.data(), tof, NB * 4, hipMemcpyDeviceToHost));
    int mism = 0, inversions = 0, worst = 0;
    for (int b = 0; b < NB; ++b) mism += hx[b] != (b & 7);
    for (int b = 8; b < NB; ++b) {                   // within an XCD (b, b - 8): did the higher block id start first?
        if (ht[b] < ht[b - 8]) { ++inversions; if (ht[b - 8] - ht[b] > worst) worst = ht[b - 8] - ht[b]; }
    }
    printf("placement: %d of %d workgroups with XCC_ID != blockIdx %% 8 (first ids:", mism, NB);
    for (int b = 0; b < 16; ++b) printf(" %d", hx[b]);
    printf("); start-order inversions inside an XCD (block b started before b - 8): %d (largest ticket gap %d)\n", inversions, worst);

    // ---- part 2
    const int groups = 128, nkt = 8, nqt = 15;
    ChainP p;
    CK(hipMalloc(&p.acc, (size_t)groups * nkt * nqt * 4096 * 4));
    CK(hipMalloc(&p.sem, groups * nqt * 4)); CK(hipMalloc(&p.err, 16)); CK(hipMalloc(&p.xcc_of, groups * 4));
    p.groups = groups; p.nkt = nkt; p.nqt = nqt;
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const int works[] = {0, 3000};
    for (int spread = 0; spread <= 1; ++spread)
        for (int wi = 0; wi < 2; ++wi)
            for (int mode = 0; mode < 6; ++mode) {
                // mode: 0 no chain | 1 ld plain st plain | 2 ld sc1 st plain | 3 ld sc0sc1 st plain | 4 ld sc0sc1 st sc0sc1 | 5 ld sc1 st sc0sc1
                p.chain = mode != 0; p.spread = spread; p.work = works[wi];
                p.ld = mode == 1 ? 0 : (mode == 2 || mode == 5) ? 1 : 2;
                p.st = mode >= 4 ? 1 : 0;
                if (spread && mode == 0) continue;
                float best = 1e9f;
                int err[4] = {0, 0, 0, 0};
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(p.sem, 0, groups * nqt * 4)); CK(hipMemset(p.err, 0, 16)); CK(hipMemset(p.xcc_of, 0, groups * 4));
                    CK(hipMemset(p.acc, 0xFF, (size_t)groups * nkt * nqt * 4096 * 4));       // NaN patterns: stale reads are visible
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(t0, 0));
                    hipLaunchKernelGGL(chain, dim3(8 * ((groups + 7) / 8) * nkt), dim3(256), 0, 0, p);
                    CK(hipEventRecord(t1, 0));
                    CK(hipDeviceSynchronize());
                    float ms;
                    CK(hipEventElapsedTime(&ms, t0, t1));
                    if (ms < best) best = ms;
                    int e[4];
                    CK(hipMemcpy(e, p.err, 16, hipMemcpyDeviceToHost));
                    for (int k = 0; k < 3; ++k) err[k] += e[k];
                }
                printf("chain spread=%d work=%d mode=%d (ld %d st %d): %.1f us, wrong values %d, spin timeouts %d, xcc mismatches %d\n",
                       spread, p.work, mode, p.ld, p.st, best * 1e3f, err[0], err[1], err[2]);
            }
    return 0;
}
