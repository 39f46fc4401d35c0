// The row-local chain kernel, SECOND FORM (round 5, late): eight waves of 16 rows per 128-row workgroup on
// v_mfma_f32_16x16x32_bf16 - 256 registers per lane, TWO waves per SIMD.  Same mathematics, interface, weight-stream size
// and stage count as dense_chain_kernel (chain.hip: read its header first); what changes is who overlaps with whom: the first
// form runs one 512-register wave per SIMD, and a single in-order wave adds its multiplies (27 % of its cycles), its 12 k
// vector instructions (28 %) and its waits (30 %) up; here the second wave of a SIMD issues while the first one's MFMA holds
// the issue port or waits for LDS.  The price: a 16-row wave multiplies a 1 KB weight fragment against half as many rows, so
// the fragment reads run at the LDS's full 256 B/clk while the matrix pipe is busy.
//
// Layout (lane = row t = lane & 15, k-group kg = lane >> 4): an accumulator tile is 16 features x 16 rows, 4 registers per
// lane = features 16 j + 4 kg + r; a 32-wide k-block of the next product is TWO consecutive tiles: slot e of lane-group kg is
// feature 16 (e >> 2) + 4 kg + (e & 3) of the block - the order the weights are packed in (dense_chain16_pack_kernel).
// Included by chain.hip (shares its helpers and parameter struct).
#pragma once

typedef float f32x4v __attribute__((ext_vector_type(4)));

// NW = waves per workgroup: 8 (128 rows, two waves per SIMD: decoder-size launches) or 4 (64 rows, one wave per SIMD, round 6: a
// workgroup's lifetime is its 52 weight stages whatever the row count - ~60 us at two waves per SIMD - so below ~16 k rows,
// where 128-row tiles leave half of the CUs without a workgroup, 64-row tiles halve each SIMD's matrix work per stage and
// double the workgroups; the four launches this kernel replaces cost ~70 us per block at 6 400 rows in the step)
#define C16_NW_MAX 8
#define C16_SLOT_BYTES (16 * CH_SLOT_LD * 2)            // 2 304: 16 rows x 64 bf16 (144-byte rows), or 16 rows x 32 fp32 (36-float rows)
#define C16_SCR_BYTES (C16_NW_MAX * C16_SLOT_BYTES)
#define C16_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// one stage: 32 fragments x one 16x16x32 multiply, four groups of eight, the read of group g + 1 behind the multiplies of g
template <class MF, class DMA>
__device__ __forceinline__ void c16_stage(const unsigned char* Fs, MF&& mf, DMA&& dma) {
    bf16x8 a0[8], a1[8];
#define C16_FRAG(g, i) (*reinterpret_cast<const bf16x8*>(Fs + ((g) * 8 + (i)) * CH_FRAG_BYTES))
#define C16_GROUP(cur, nxt, g)                                                       \
    __builtin_amdgcn_sched_barrier(0);                                               \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                  \
        mf(g, i, cur[i]);                                                            \
        nxt[i] = C16_FRAG((g) + 1, i);                                               \
    }                                                                                \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                           \
    }                                                                                \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) a0[i] = C16_FRAG(0, i);
    C16_GROUP(a0, a1, 0)
    C16_GROUP(a1, a0, 1)
    dma(0);
    C16_GROUP(a0, a1, 2)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) mf(3, i, a1[i]);
    __builtin_amdgcn_sched_barrier(0);
    dma(1);
#undef C16_GROUP
#undef C16_FRAG
}

// the wave's 16 rows x 64 features (four 16-feature tiles, bf16) -> its scratch slot -> global rows, 16 bytes per lane
__device__ __forceinline__ void c16_slot_write(unsigned char* slot, int t, int kg, int u, uint2 v) {
    *reinterpret_cast<uint2*>(slot + (t * CH_SLOT_LD + 16 * u + 4 * kg) * 2) = v;
}
__device__ __forceinline__ void c16_slot_flush(const unsigned char* slot, uint16_t* dst, long ld, int col0, int row0, int M, int lane,
                                               uint32_t* bits, int bits_wide, int nbchunk) {
    const int r8 = lane >> 3, c8 = (lane & 7) * 8;
    uint4 v[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) v[it] = *reinterpret_cast<const uint4*>(slot + ((r8 + 8 * it) * CH_SLOT_LD + c8) * 2);
    uint16_t* d0 = dst + (long)(row0 + r8) * ld + col0 + c8;
    if (row0 + 16 <= M) {
#pragma unroll
        for (int it = 0; it < 2; ++it) *reinterpret_cast<uint4*>(d0 + (long)(8 * it) * ld) = v[it];
    } else {
#pragma unroll
        for (int it = 0; it < 2; ++it)
            if (row0 + r8 + 8 * it < M) *reinterpret_cast<uint4*>(d0 + (long)(8 * it) * ld) = v[it];
    }
    if (bits != nullptr && row0 < M) {
        // rows of the 64-row tile: R = (row0 & 63) + r8 + 8 it, row0 & 63 = 16 q (q = the wave's quarter of the tile)
        const uint32_t b0 = ch_pos_bits(v[0]), b1 = ch_pos_bits(v[1]);
        const long tile = row0 >> 6;
        const int q = (row0 >> 4) & 3;
        unsigned char* bytes = reinterpret_cast<unsigned char*>(bits);
        if (bits_wide) {
            // gemm_k256_wide_kernel: thread = (R & 7) * 32 + column / 8 of a 256-column block, byte R >> 3 of its 8-byte word
            const int chunk = col0 >> 8, cb = ((col0 & 255) >> 3) + (lane & 7);
            const long word = (tile * nbchunk + chunk) * 256 + r8 * 32 + cb;
            *reinterpret_cast<uint16_t*>(bytes + word * 8 + 2 * q) = (uint16_t)(b0 | (b1 << 8));
        } else {
            // gemm_k256_kernel: thread = (R & 15) * 16 + column / 8 of a 128-column block, byte R >> 4 of its 4-byte word
            const int chunk = col0 >> 7, cb = ((col0 & 127) >> 3) + (lane & 7);
            const long base = (tile * nbchunk + chunk) * 256;
            bytes[(base + r8 * 16 + cb) * 4 + q] = (unsigned char)b0;
            bytes[(base + (r8 + 8) * 16 + cb) * 4 + q] = (unsigned char)b1;
        }
    }
}
__device__ __forceinline__ void c16_slot_flush_f32(const unsigned char* slot, float* dst, int col0, int row0, int M, int lane) {
    const int r8 = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int r = r8 + 8 * it;
        const float4 v = *reinterpret_cast<const float4*>(slot + (r * 36 + c4) * 4);
        if (row0 + r < M) *reinterpret_cast<float4*>(dst + (long)(row0 + r) * CH_D + col0 + c4) = v;
    }
}

__device__ __forceinline__ void c16_bias_init(f32x4v (&Z)[16], const float* bias, int kg) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float4 b4 = *reinterpret_cast<const float4*>(bias + 16 * j + 4 * kg);
        Z[j][0] = b4.x; Z[j][1] = b4.y; Z[j][2] = b4.z; Z[j][3] = b4.w;
    }
}

// LayerNorm of the wave's 16 rows in the accumulator layout (Z holds product + bias); R: the residual as bf16 fragments
// (fragment kb = tiles 2 kb, 2 kb + 1); Y: the result's fragments (may alias R).
template <bool Y32>
__device__ __forceinline__ void c16_layernorm(f32x4v (&Z)[16], const bf16x8 (&R)[8], bf16x8 (&Y)[8], const ChainP& p, const float* gamma,
                                              const float* beta, uint64_t drop_base, uint32_t site, int row, int rowc, int row0, bool padded,
                                              unsigned char* slot, int lane, uint16_t* y_bf, uint16_t* xhat, float* rstd_out, float* y32,
                                              bool wr = true) {
    // wr (wave-uniform): false = compute the result's fragments only (the SPLIT form's second workgroup: its partner stores)
    const int t = lane & 15, kg = lane >> 4;
    const uint64_t key = p.thr ? ttsmi_drop_key_of(drop_base, site) : 0;
    const uint32_t rb = ttsmi_row_base(key, (uint32_t)rowc);
    float sum4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c0 = 16 * j + 4 * kg;
        float v[4] = {Z[j][0], Z[j][1], Z[j][2], Z[j][3]};
        if (p.thr) {
            const uint32_t h0 = ttsmi_pair_hash(rb, (uint32_t)c0), h1 = ttsmi_pair_hash(rb, (uint32_t)(c0 + 2));
            v[0] *= ((h0 & 0xFFFFu) >= p.thr) ? p.inv_keep : 0.f;
            v[1] *= ((h0 >> 16) >= p.thr) ? p.inv_keep : 0.f;
            v[2] *= ((h1 & 0xFFFFu) >= p.thr) ? p.inv_keep : 0.f;
            v[3] *= ((h1 >> 16) >= p.thr) ? p.inv_keep : 0.f;
        }
        const bf16x8& rr = R[j >> 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] += ch_bf(rr, 4 * (j & 1) + e);
            Z[j][e] = v[e];
            sum4[e] += v[e];
        }
    }
    const float invC = 1.0f / (float)CH_D;
    float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * invC;
    float q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = Z[j][e] - mean;
            q4[e] += v * v;
        }
    float q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rstd = __builtin_amdgcn_rsqf(q * invC + p.eps);
    const float nmr = -mean * rstd;
    if (kg == 0 && row < p.M && wr) rstd_out[row] = rstd;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {                 // 64 features (four tiles) per round through the wave's scratch slot
        uint2 xh_q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = 4 * cc + u, c0 = 16 * j + 4 * kg;
            const float4 gm = *reinterpret_cast<const float4*>(gamma + c0);
            const float4 bt = *reinterpret_cast<const float4*>(beta + c0);
            const float xh[4] = {fmaf(Z[j][0], rstd, nmr), fmaf(Z[j][1], rstd, nmr), fmaf(Z[j][2], rstd, nmr), fmaf(Z[j][3], rstd, nmr)};
            float y[4] = {fmaf(xh[0], gm.x, bt.x), fmaf(xh[1], gm.y, bt.y), fmaf(xh[2], gm.z, bt.z), fmaf(xh[3], gm.w, bt.w)};
            if (padded) { y[0] = 0.f; y[1] = 0.f; y[2] = 0.f; y[3] = 0.f; }
            xh_q[u] = ch_pack4(xh[0], xh[1], xh[2], xh[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Y[j >> 1][4 * (j & 1) + e] = (__bf16)y[e];
                if constexpr (Y32) Z[j][e] = y[e];
            }
        }
        if (!wr) continue;
#pragma unroll
        for (int u = 0; u < 4; ++u) c16_slot_write(slot, t, kg, u, xh_q[u]);
        ch_lds_fence();
        c16_slot_flush(slot, xhat, CH_D, 64 * cc, row0, p.M, lane, nullptr, 0, 0);
        ch_lds_fence();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = 4 * cc + u;
            const bf16x8& yy = Y[j >> 1];
            bf16x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = yy[4 * (j & 1) + e];
            c16_slot_write(slot, t, kg, u, *reinterpret_cast<uint2*>(&h));
        }
        ch_lds_fence();
        c16_slot_flush(slot, y_bf, CH_D, 64 * cc, row0, p.M, lane, nullptr, 0, 0);
        ch_lds_fence();
        if (Y32 && y32 != nullptr) {
#pragma unroll
            for (int hv = 0; hv < 2; ++hv) {         // 32 features (two tiles) per fp32 round
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int j = 4 * cc + 2 * hv + u;
                    *reinterpret_cast<float4*>(slot + (t * 36 + 16 * u + 4 * kg) * 4) = make_float4(Z[j][0], Z[j][1], Z[j][2], Z[j][3]);
                }
                ch_lds_fence();
                c16_slot_flush_f32(slot, y32, 64 * cc + 32 * hv, row0, p.M, lane);
                ch_lds_fence();
            }
        }
    }
}

// one DMA call of a wave = HALF of its pieces of a stage (NDMA = 32 / NW pieces of 1 KB per wave and stage; `half` 0 / 1), issued
// with one m0 set-up and instruction offsets: 2 pieces (NW = 8) or 4 (NW = 4)
template <int NDMA>
__device__ __forceinline__ void c16_issue_half(const unsigned char* src, unsigned dst, int half) {
    static_assert(NDMA == 4 || NDMA == 8, "8 or 4 waves");
    src += half * (NDMA / 2) * CH_FRAG_BYTES;
    dst += (unsigned)(half * (NDMA / 2) * CH_FRAG_BYTES);
    if constexpr (NDMA == 4)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024"
                     ::"v"(src), "s"(dst) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                     ::"v"(src), "s"(dst) : "memory", "m0");
}

// LOADER waves (round 6, the four-wave form): with one compute wave per SIMD nothing hides the issue time of a stage's DMA
// pieces - eight per wave, ~400 of a stage's ~1 300 cycles (profiles/r06_chain64_phases.txt) - so NL = 2 extra waves do nothing
// but stream the weights: each issues 16 of a stage's 32 pieces right after the stage barrier that retires the ring slot, waits
// for its own pieces with a counted vmcnt and joins every workgroup barrier the compute waves execute (`extra_barrier_after`:
// the backward kernel has one more, behind the barrier of that stage).  The compute waves then neither issue nor wait for DMA.
struct C16IdentityStage { __device__ __forceinline__ int operator()(int s) const { return s; } };
template <int NL, class MAP = C16IdentityStage>
__device__ __forceinline__ void c16_loader_loop(const unsigned char* wpack, unsigned ring_off, int nst, int lw, int lane, int extra_barrier_after,
                                                MAP packed_stage = MAP()) {
    constexpr int PL = CH_STAGE_FRAGS / NL;                      // pieces per loader wave and stage
    static_assert(PL % 4 == 0 && 3 * PL <= 63, "vmcnt is a 6-bit counter");
    const unsigned char* src0 = wpack + (size_t)lw * PL * CH_FRAG_BYTES + lane * 16;
    const unsigned dst0 = ring_off + (unsigned)lw * PL * CH_FRAG_BYTES;
    auto issue = [&](int s) {
        if (s >= nst) return;
        const unsigned char* src = src0 + (size_t)packed_stage(s) * CH_STAGE_BYTES;    // (the workgroup's stage s of the packed stream)
        const unsigned dst = __builtin_amdgcn_readfirstlane(dst0 + (unsigned)(s % CH_NRING) * CH_STAGE_BYTES);
#pragma unroll
        for (int k = 0; k < PL / 4; ++k) c16_issue_half<8>(src + k * 4 * CH_FRAG_BYTES, dst + (unsigned)(k * 4 * CH_FRAG_BYTES), 0);
    };
#pragma unroll
    for (int s = 0; s < CH_NRING - 1; ++s) issue(s);
    for (int s = 0; s < nst; ++s) {
        if (s + 2 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PL) : "memory");
        else if (s + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ch_barrier();
        if (s == extra_barrier_after) ch_barrier();
        issue(s + CH_NRING - 1);
    }
}

// coherent accesses of the SPLIT form's exchange (sc0 sc1: system scope - the partner workgroup may sit on another XCD, whose L2
// is not coherent with this one's)
// (OFF: the instruction's immediate byte offset, 0 .. 4095 - one address register serves four of a wave's 1 KB rows)
template <int OFF>
__device__ __forceinline__ void c16_store16_sc(float* ptr, f32x4v v) {
    asm volatile("global_store_dwordx4 %0, %1, off offset:%2 sc0 sc1" ::"v"(ptr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void c16_load16_sc(f32x4v& v, const float* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc0 sc1" : "=v"(v) : "v"(ptr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void c16_store_flag_sc(uint32_t* ptr, uint32_t v) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(ptr), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t c16_load_flag_sc(const uint32_t* ptr) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}

// wait for the partner's flag.  A partner that never raises it (it was never dispatched: something else holds every CU for good)
// would hang the queue and the device with it: after ~4 M polls (seconds) the wave traps instead - the launch fails loudly.
// (Two streams launching split forms at the same time - 2 x 208 workgroups for 256 CUs - make progress: workgroups are dispatched
// in order, so at most one pair per XCD is half-resident; tools/probe_chain_split_two_streams.py, profiles/r06_chain_split_two_streams.txt.)
// Measured and not kept: an XCD-aware exchange (waves publish their XCC id; partners under one L2 store and load at agent scope):
// 39.9 against 37.5 us forward - the system-scope round trips are not what the exchange costs (profiles/r06_chain_split_xcd_exchange_ab.txt).
__device__ __forceinline__ void c16_wait_flag(const uint32_t* flag) {
    int polls = 0;
    while (__builtin_amdgcn_readfirstlane(c16_load_flag_sc(flag)) != 1u) {
        __builtin_amdgcn_s_sleep(2);
        if (++polls > (1 << 22)) __builtin_trap();
    }
}

// SPLIT (round 6, row counts up to 8 192): TWO workgroups per 64-row tile.  Both run the o-projection and res-norm 1, each
// streams HALF of the FFN's hidden chunks and half of the qkv columns - 30 weight stages instead of 52 - and they exchange the
// fp32 partial sum of the FFN output (wave by wave: a wave's 16 rows only need the same wave of the partner) through global
// memory between the FFN and res-norm 2.  Workgroups b and b + 8 of a group of 16 share a tile.  Both compute P0 + P1 in this
// order, so they normalise identical sums; the first of the pair stores the row-wise results.  A launch's time below 16 k rows is
// its stages whatever the row count - this form halves them where the CUs to run twice the workgroups are idle anyway.
template <bool Y32, int NW, int NL = 0, bool SPLIT = false>
__global__ __launch_bounds__((NW + NL) * 64, 1) __attribute__((amdgpu_waves_per_eu((NW + 3) / 4, (NW + NL + 3) / 4))) void dense_chain16_kernel(ChainP p) {
    static_assert(!SPLIT || NL > 0, "the split form streams through loader waves");
    constexpr int C16_ROWS = NW * 16, C16_NDMA = CH_STAGE_FRAGS / NW;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[CH_NRING * CH_STAGE_BYTES + C16_SCR_BYTES + CH_PAR_FLOATS * 4];
    unsigned char* scr = smem + CH_NRING * CH_STAGE_BYTES;
    float* par = reinterpret_cast<float*>(scr + C16_SCR_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, t = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int side = SPLIT ? (blockIdx.x >> 3) & 1 : 0;                                  // which of the tile's two workgroups
    const int tile = SPLIT ? (blockIdx.x >> 4) * 8 + (blockIdx.x & 7) : blockIdx.x;
    const int m0 = tile * C16_ROWS;
    const int row0 = m0 + wave * 16, row = row0 + t, rowc = min(row, p.M - 1);
    const int nc_w = SPLIT ? p.nchunk / 2 : p.nchunk, c_first = side * nc_w;             // this workgroup's hidden chunks
    const int nq_w = p.qkv == nullptr ? 0 : SPLIT ? CH_QKV_STAGES / 2 : CH_QKV_STAGES;   // ... and qkv stages
    const int nst = SPLIT ? CH_WO_STAGES + 2 * nc_w + nq_w : p.nstages;
    // the workgroup's stage s in the packed stream: three contiguous runs (o-projection, its chunks, its qkv columns)
    auto packed_stage = [=](int s) -> int {
        if (!SPLIT || s < CH_WO_STAGES) return s;
        if (s < CH_WO_STAGES + 2 * nc_w) return s + 2 * c_first;
        return s + 2 * (p.nchunk - nc_w) + side * nq_w;
    };
    const unsigned ring_off = ch_lds_offset(smem);
#ifdef TTSMI_ABLATION_BUILD
    unsigned long long tph[8], twait = 0;
    int nph = 0;
#define C16_STAMP() tph[nph++] = __builtin_readcyclecounter()
#else
#define C16_STAMP()
#endif
    C16_STAMP();

    // two of this wave's four pieces of stage s (one m0 set-up, instruction offsets on both addresses)
    const unsigned char* wsrc = p.wpack + (size_t)wave * C16_NDMA * CH_FRAG_BYTES + lane * 16;
    const unsigned wdst = ring_off + (unsigned)wave * C16_NDMA * CH_FRAG_BYTES;
    if constexpr (NL > 0) {
        if (wave >= NW) {                                         // (wave-uniform: the loader waves never touch rows)
            c16_loader_loop<NL>(p.wpack, ring_off, nst, wave - NW, lane, -1, packed_stage);
            return;
        }
    }
    auto issue2 = [&](int s, int g) {
        if constexpr (NL == 0) {                                  // (NL > 0: the loader waves stream the weights)
        if (s >= nst) return;
        c16_issue_half<C16_NDMA>(wsrc + (size_t)s * CH_STAGE_BYTES, __builtin_amdgcn_readfirstlane(wdst + (unsigned)(s % CH_NRING) * CH_STAGE_BYTES), g & 1);
        }
    };
    // stage s has landed once at most the pieces of the two stages behind it are outstanding on every wave (chain.hip)
    auto stage_begin = [&](int s) -> const unsigned char* {
#ifdef TTSMI_ABLATION_BUILD
        const unsigned long long tw0 = __builtin_readcyclecounter();
#endif
        if constexpr (NL == 0) {
            if (s + 2 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C16_NDMA) : "memory");
            else if (s + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C16_NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ch_barrier();
#ifdef TTSMI_ABLATION_BUILD
        twait += __builtin_readcyclecounter() - tw0;
#endif
        return smem + (s % CH_NRING) * CH_STAGE_BYTES + lane * 16;
    };

    // ---- the wave's 16 rows of [h | ctx] as B fragments (k-block kb: features 32 kb + 16 (e >> 2) + 4 kg + (e & 3)), read
    // with full-line accesses and turned into fragments through the wave's 4 KB of ring slot 3 (chunk c of row r at c ^ (r & 15))
    bf16x8 XH[8], XC[8];
    {
        unsigned char* xs = smem + (CH_NRING - 1) * CH_STAGE_BYTES + wave * 4096;
        const int lr = lane >> 4, lc = lane & 15;
        uint4 raw0[4], raw1[4], raw2[4], raw3[4];
#define C16_XLOAD(dst, base, half)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                    \
        const int r = row0 + 4 * i + lr;                                                                               \
        dst[i] = r < p.M ? *reinterpret_cast<const uint4*>((base) + (long)r * CH_D + (half) * 128 + lc * 8)             \
                         : make_uint4(0u, 0u, 0u, 0u);                                                                  \
    }
        C16_XLOAD(raw0, p.h_bf, 0)
        C16_XLOAD(raw1, p.h_bf, 1)
        C16_XLOAD(raw2, p.cx, 0)
        C16_XLOAD(raw3, p.cx, 1)
#undef C16_XLOAD
#define C16_XFRAGS(src, dstarr, base)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                    \
        const int r = 4 * i + lr;                                                                                      \
        *reinterpret_cast<uint4*>(xs + r * 256 + ((lc ^ (r & 15)) << 4)) = src[i];                                     \
    }                                                                                                                  \
    ch_lds_fence();                                                                                                    \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                    \
        const int c = 4 * q + (kg >> 1);                                                                               \
        const uint2 lo = *reinterpret_cast<const uint2*>(xs + t * 256 + ((c ^ t) << 4) + 8 * (kg & 1));                \
        const uint2 hi = *reinterpret_cast<const uint2*>(xs + t * 256 + (((c + 2) ^ t) << 4) + 8 * (kg & 1));          \
        const uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);                                                            \
        dstarr[(base) + q] = *reinterpret_cast<const bf16x8*>(&v);                                                     \
    }                                                                                                                  \
    ch_lds_fence();
        C16_XFRAGS(raw0, XH, 0)
        C16_XFRAGS(raw1, XH, 4)
        C16_XFRAGS(raw2, XC, 0)
        C16_XFRAGS(raw3, XC, 4)
#undef C16_XFRAGS
    }
    const bool padded = p.row_pad != nullptr && p.row_pad[rowc] != 0;
    const uint64_t drop_base = p.thr ? ttsmi_drop_base(p.seed, p.step_dev) : 0;      // (read here, in front of the first DMA issue)
    {
        auto stage_vec = [&](const float* src, int off, int n) {
            for (int i = tid * 4; i < n; i += NW * 64 * 4) *reinterpret_cast<float4*>(par + off + i) = *reinterpret_cast<const float4*>(src + i);
        };
        stage_vec(p.bo, CH_P_BO, CH_D); stage_vec(p.ln1_g, CH_P_G1, CH_D); stage_vec(p.ln1_b, CH_P_BE1, CH_D);
        stage_vec(p.b2, CH_P_B2, CH_D); stage_vec(p.ln2_g, CH_P_G2, CH_D); stage_vec(p.ln2_b, CH_P_BE2, CH_D);
        stage_vec(p.b1, CH_P_B1, p.F);
        if (p.qkv != nullptr) stage_vec(p.bqkv, CH_P_BQ, 3 * CH_D);
    }
#pragma unroll
    for (int s = 0; s < CH_NRING - 1; ++s) {
        issue2(s, 0);
        issue2(s, 1);
    }

    f32x4v Z[16];
    int S = 0;
    C16_STAMP();
    unsigned char* slot = scr + wave * C16_SLOT_BYTES;
    bf16x8(&Y)[8] = XH;
    const int nbchunk = p.bits_wide ? p.F / 256 : p.F / 128;
    for (int half = 0; half < p.nhalf; ++half) {
        if (half == 0) {
            // o-projection: 8 stages of (2 k-blocks x 16 output tiles)
#pragma unroll
            for (int s = 0; s < CH_WO_STAGES; ++s) {
                const unsigned char* Fs = stage_begin(S);
                if (s == 0) c16_bias_init(Z, par + CH_P_BO, kg);
                c16_stage(Fs, [&](int g, int i, const bf16x8& a) {
                              const int kb = 2 * s + (g >> 1), j = (g & 1) * 8 + i;
                              Z[j] = C16_MFMA(a, (kb < 8 ? XH[kb & 7] : XC[kb & 7]), Z[j]);
                          },
                          [&](int g) { issue2(S + CH_NRING - 1, g); });
                ++S;
            }
        } else {
            // FFN: per 64 hidden features a stage of a . W1 (4 tiles x 8 k-blocks) and one of h1 . W2 (2 k-blocks x 16 tiles)
            if (side == 0) {
                c16_bias_init(Z, par + CH_P_B2, kg);
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) Z[j][e] = 0.f;
            }
            for (int cc_ = 0; cc_ < nc_w; ++cc_) {
                const int c = c_first + cc_;
                const unsigned char* Fs = stage_begin(S);
                f32x4v H[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 b4 = *reinterpret_cast<const float4*>(par + CH_P_B1 + 64 * c + 16 * u + 4 * kg);
                    H[u][0] = b4.x; H[u][1] = b4.y; H[u][2] = b4.z; H[u][3] = b4.w;
                }
                c16_stage(Fs, [&](int g, int i, const bf16x8& a) { H[i & 3] = C16_MFMA(a, Y[2 * g + (i >> 2)], H[i & 3]); },
                          [&](int g) { issue2(S + CH_NRING - 1, g); });
                ++S;
                bf16x8 hf[2];
                uint32_t lane_bits = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    uint2 h = ch_pack4(H[u][0], H[u][1], H[u][2], H[u][3]);
                    h.x = ch_relu2(h.x);
                    h.y = ch_relu2(h.y);
                    const bf16x4 hb = *reinterpret_cast<const bf16x4*>(&h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) hf[u >> 1][4 * (u & 1) + e] = hb[e];
                    c16_slot_write(slot, t, kg, u, h);
                    // (a positive bf16 after the ReLU = a non-zero half: bit 4 u + e of the lane's word)
                    lane_bits |= (((h.x & 0xFFFFu) ? 1u : 0u) | ((h.x >> 16) ? 2u : 0u) | ((h.y & 0xFFFFu) ? 4u : 0u) | ((h.y >> 16) ? 8u : 0u)) << (4 * u);
                }
                // the ReLU pattern in the BACKWARD CHAIN's layout (chain16b.h): one 16-bit word per lane, 128 bytes per wave and chunk
                if (p.bits_wide == 2 && p.relu_bits != nullptr && row0 < p.M)
                    reinterpret_cast<uint16_t*>(p.relu_bits)[((long)(row0 >> 4) * p.nchunk + c) * 64 + lane] = (uint16_t)lane_bits;
                Fs = stage_begin(S);
                c16_slot_flush(slot, p.h1, p.F, 64 * c, row0, p.M, lane, p.bits_wide == 2 ? nullptr : p.relu_bits, p.bits_wide, nbchunk);
                c16_stage(Fs, [&](int g, int i, const bf16x8& a) { const int j = (g & 1) * 8 + i; Z[j] = C16_MFMA(a, hf[g >> 1], Z[j]); },
                          [&](int g) { issue2(S + CH_NRING - 1, g); });
                ++S;
            }
            ch_lds_fence();
            if constexpr (SPLIT) {
                // ---- the partner's partial sum: store mine, raise my flag, wait for the partner's, load, lower it (its only reader)
                const long wslot = ((long)tile * 2 + side) * NW + wave, oslot = ((long)tile * 2 + (side ^ 1)) * NW + wave;
                float* mine = p.xbuf + wslot * (16 * 256) + lane * 4;
                const float* theirs = p.xbuf + oslot * (16 * 256) + lane * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float* m4 = mine + q * 1024;
                    c16_store16_sc<0>(m4, Z[4 * q]); c16_store16_sc<1024>(m4, Z[4 * q + 1]);
                    c16_store16_sc<2048>(m4, Z[4 * q + 2]); c16_store16_sc<3072>(m4, Z[4 * q + 3]);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) c16_store_flag_sc(p.xflag + wslot, 1u);
                c16_wait_flag(p.xflag + oslot);
                f32x4v O[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float* t4 = theirs + q * 1024;
                    c16_load16_sc<0>(O[4 * q], t4); c16_load16_sc<1024>(O[4 * q + 1], t4);
                    c16_load16_sc<2048>(O[4 * q + 2], t4); c16_load16_sc<3072>(O[4 * q + 3], t4);
                }
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(O[0]), "+v"(O[1]), "+v"(O[2]), "+v"(O[3]), "+v"(O[4]), "+v"(O[5]), "+v"(O[6]), "+v"(O[7]), "+v"(O[8]), "+v"(O[9]),
                               "+v"(O[10]), "+v"(O[11]), "+v"(O[12]), "+v"(O[13]), "+v"(O[14]), "+v"(O[15])
                             :
                             : "memory");
#pragma unroll
                for (int jj = 0; jj < 16; ++jj)
#pragma unroll
                    for (int e = 0; e < 4; ++e) Z[jj][e] = side == 0 ? Z[jj][e] + O[jj][e] : O[jj][e] + Z[jj][e];     // P0 + P1 on both sides
                if (lane == 0) c16_store_flag_sc(p.xflag + oslot, 0u);
            }
        }
        C16_STAMP();
        c16_layernorm<Y32>(Z, Y, Y, p, par + (half ? CH_P_G2 : CH_P_G1), par + (half ? CH_P_BE2 : CH_P_BE1), drop_base, half ? p.site_ln2 : p.site_ln1, row,
                           rowc, row0, padded, slot, lane, half ? p.out_bf : p.a_bf, half ? p.xhat2 : p.xhat1, half ? p.rstd2 : p.rstd1,
                           half ? p.out32 : nullptr, side == 0);
        C16_STAMP();
    }

    // the next block's qkv projection: 12 stages of (4 output tiles x 8 k-blocks)
    if (p.qkv != nullptr) {
        const int q0 = side * nq_w;                                 // (SPLIT: this workgroup's half of the 768 columns)
        for (int s_ = 0; s_ < nq_w; ++s_) {
            const int s = q0 + s_;
            const unsigned char* Fs = stage_begin(S);
            if (s_ > 0) c16_slot_flush(slot, p.qkv, 3 * CH_D, 64 * (s - 1), row0, p.M, lane, nullptr, 0, 0);
            f32x4v acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 b4 = *reinterpret_cast<const float4*>(par + CH_P_BQ + 64 * s + 16 * u + 4 * kg);
                acc[u][0] = b4.x; acc[u][1] = b4.y; acc[u][2] = b4.z; acc[u][3] = b4.w;
            }
            c16_stage(Fs, [&](int g, int i, const bf16x8& a) { acc[i & 3] = C16_MFMA(a, Y[2 * g + (i >> 2)], acc[i & 3]); },
                      [&](int g) { issue2(S + CH_NRING - 1, g); });
            ++S;
            ch_lds_fence();
#pragma unroll
            for (int u = 0; u < 4; ++u) c16_slot_write(slot, t, kg, u, ch_pack4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]));
        }
        ch_lds_fence();
        c16_slot_flush(slot, p.qkv, 3 * CH_D, 64 * (q0 + nq_w - 1), row0, p.M, lane, nullptr, 0, 0);
    }
#ifdef TTSMI_ABLATION_BUILD
    C16_STAMP();
    if (p.dbg && lane == 0) {              // [workgroup][wave][8]: start, prologue, o-projection, LN1, FFN, LN2, qkv, time in stage waits
        unsigned long long* o = p.dbg + ((long)blockIdx.x * NW + wave) * 8;
        o[0] = tph[0];
        for (int i = 1; i < 7; ++i) o[i] = tph[i] - tph[i - 1];
        o[7] = twait;
    }
#endif
}

// the weight stream of this form: fragment = 16 output features x one 32-wide k-block, slot (kg, e) = k 16 (e >> 2) + 4 kg + (e & 3)
__device__ __forceinline__ void c16_pack_body(const ChainPackP& p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)p.nstages * CH_STAGE_FRAGS * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63), f = (int)((idx >> 6) & 31), S = (int)(idx >> 11);
    const int n16 = lane & 15, kg = lane >> 4;
    const uint16_t* src;
    long ld;
    int n, kbase;
    const int wo = p.wo_stages;
    if (S < wo) {
        const int kb = f >> 4, j = f & 15;
        src = p.wo_t; ld = 2 * CH_D; n = 16 * j + n16; kbase = 32 * (2 * S + kb);
    } else if (S < wo + 2 * p.nchunk) {
        const int tt = S - wo, c = tt >> 1;
        if ((tt & 1) == 0) {
            const int q = f >> 2, u = f & 3;
            src = p.w1_t; ld = CH_D; n = 64 * c + 16 * u + n16; kbase = 32 * q;
        } else {
            const int kb = f >> 4, j = f & 15;
            src = p.w2_t; ld = p.F; n = 16 * j + n16; kbase = 64 * c + 32 * kb;
        }
    } else {
        const int s = S - wo - 2 * p.nchunk, q = f >> 2, u = f & 3;
        src = p.wqkv_t; ld = CH_D; n = 64 * s + 16 * u + n16; kbase = 32 * q;
    }
    const uint16_t* r = src + (long)n * ld + kbase + 4 * kg;
    const uint2 lo = *reinterpret_cast<const uint2*>(r), hi = *reinterpret_cast<const uint2*>(r + 16);
    *reinterpret_cast<uint4*>(p.out + idx * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
}
__global__ __launch_bounds__(256) void dense_chain16_pack_kernel(ChainPackP p) { c16_pack_body(p); }
// several weight streams in one launch (a train step repacks 24 of them after every optimiser update): blockIdx.y = job
#define C16_PACK_MAX_JOBS 32
struct ChainPackJobs { ChainPackP job[C16_PACK_MAX_JOBS]; };
__global__ __launch_bounds__(256) void dense_chain16_pack_jobs_kernel(ChainPackJobs J) { c16_pack_body(J.job[blockIdx.y]); }
