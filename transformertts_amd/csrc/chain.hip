// The row-local chain of a dense block as ONE persistent-per-tile kernel (TTSMI_BF16 path, d_model = 256):
//
//   a   = LN1( keep([h | ctx] . Wo + bo) + h ) * rowmask                      (model/layers.py:148-150, 211, 229)
//   h1  = relu(a . W1 + b1)                                                   (model/layers.py:99)
//   out = LN2( keep(h1 . W2 + b2) + a ) * rowmask                             (model/layers.py:100-102, 230)
//   qkv'= out . Wqkv' + bqkv'      (the NEXT block's projection, model/layers.py:116-118 - when there is a next block)
//
// Until round 4 these were four launches (full-row GEMM + LN, K = 256 GEMM, full-row GEMM + LN, K = 256 GEMM), each
// re-reading from HBM what the previous one had just written: a_bf twice, h1 once, out_bf once (118 MB per decoder block
// at M = 28 800) and each paying its own launch ramp, epilogue and tail.  Here a workgroup takes 128 rows through the whole
// chain; nothing but the tensors the BACKWARD needs (a_bf, x^1, rstd1, h1, x^2, rstd2, out_bf, qkv') is written, nothing
// is read back.
//
// Layout: a wave owns 32 rows.  Every product is computed transposed (C^T = W . X^T: the MFMA A operand is a 32 x 16
// weight fragment, the B operand the activation fragment), so a lane holds ONE row (lane & 31) and 16 of every 32 output
// features in its accumulator registers - and an accumulator register block IS the B operand of the next product:
// registers 8p .. 8p + 7 of output tile j, rounded to bf16, are the fragment of k-group 2j + p (the register-feedback
// trick of the attention kernels, attention_bf16.hip:to_frags).  The k order inside such a group is the accumulator's
// (features 16q + {0..3, 8..11} + 4 * (lane >> 5)), so the weights are stored pre-permuted to match: ttsmi_dense_chain_pack
// writes the four matrices of a chain as ONE linear stream of 1 KB MFMA A fragments in exactly the order the kernel
// multiplies them (52 stages of 32 fragments at F = 1 024; 1.66 MB per block and step, repacked after every Adam step).
// A row's LayerNorm statistics are 128 local adds and one exchange with lane ^ 32; residuals never leave registers.
//
// The weight stream arrives by LDS-DMA (global_load_lds_dwordx4, lane-linear = fragment order: no swizzle, conflict-free
// ds_read_b128) into a FOUR-stage ring of 32 KB stages: one barrier per stage, three stages in flight.  Every wave issues
// its quarter of a stage (8 pieces of 1 KB), two pieces behind each group of eight MFMAs; inside a stage every multiply is
// followed by the read of the fragment eight multiplies on.  vmcnt counts loads AND stores on gfx950 and stores are not
// ordered against loads, so the counted wait in front of a stage allows exactly the DMA pieces that are YOUNGER than the
// stage's own (16, 8 or 0): loads complete in order, outstanding stores can only make that wait longer, never shorter.
// Bias / gamma / beta vectors are staged in LDS once per workgroup; accumulators start at their bias.
//
// TWO FORMS of the forward kernel live behind ttsmi_dense_chain_fwd (TTSMI_DENSE_CHAIN_FORM, default 16):
//  * this file's dense_chain_kernel: four 32-row waves on v_mfma_f32_32x32x16_bf16, 450-512 registers, ONE wave per SIMD.
//    Correct and tested, but only level with the four launches (decoder size, 28 800 rows: 98-101 us against 105-107): with
//    a single in-order wave per SIMD the MFMA issue (27 % of the cycles), 12 k vector instructions (28 %) and the waits
//    (30 %) simply ADD - stages of 2 200-2 900 cycles for 1 024 of multiplies, while the same stage skeleton alone
//    (tools/probes/mfma_lds_overlap_probe.hip) runs at 1 336.  Six builds were measured (DESIGN.md section 4, round 5;
//    profiles/r05_chain_*.txt, r05_sq_counters_chain_third_build.txt); this is the fifth.
//  * chain16.h's dense_chain16_kernel (the default): eight 16-row waves on v_mfma_f32_16x16x32_bf16, <= 256 registers, TWO
//    waves per SIMD - what the first form's analysis asked for.  85.7 us at 28 800 rows; the train step 4.99 -> 4.85 ms.
// chain16b.h holds the BACKWARD chain on the second form's layout (ttsmi_dense_chain_bwd; step 4.85 -> 4.78 ms).
// The host uses the chains from 16 384 rows on (ops.CHAIN_MIN_ROWS): a workgroup's own latency is ~60 us whatever the row
// count, which the four launches beat at encoder sizes (57 us).
//
// LDS: 128 KB ring + 18 KB transposing scratch (a slot per wave) + 13 KB parameters = 159 KB; 512 registers per lane (one
// wave per SIMD).
#include <stdlib.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define CH_D 256
#define CH_FRAG_BYTES 1024
#define CH_STAGE_FRAGS 32
#define CH_STAGE_BYTES (CH_STAGE_FRAGS * CH_FRAG_BYTES)
#define CH_NRING 4
#define CH_NW 4
#define CH_ROWS (CH_NW * 32)
#define CH_SLOT_LD 72                                   // bf16 elements per scratch row: 64 + 8 (144-byte rows)
#define CH_SLOT_BYTES (32 * CH_SLOT_LD * 2)             // 4 608: 32 rows x 64 bf16, or 32 rows x 32 fp32 (36-float rows)
#define CH_SCR_BYTES (CH_NW * CH_SLOT_BYTES)            // a private slot per wave
#define CH_NDMA (CH_STAGE_FRAGS / CH_NW)                // DMA pieces per wave and stage: 8
#define CH_MAXF 1024                                    // the staged parameter block holds b1 up to this many features
#define CH_PAR_FLOATS (6 * CH_D + CH_MAXF + 3 * CH_D)   // bo g1 be1 b2 g2 be2 | b1 | bqkv
#define CH_P_BO 0
#define CH_P_G1 (1 * CH_D)
#define CH_P_BE1 (2 * CH_D)
#define CH_P_B2 (3 * CH_D)
#define CH_P_G2 (4 * CH_D)
#define CH_P_BE2 (5 * CH_D)
#define CH_P_B1 (6 * CH_D)
#define CH_P_BQ (6 * CH_D + CH_MAXF)
#define CH_WO_STAGES 8                                  // K = 512 in steps of 64
#define CH_QKV_STAGES 12                                // 768 output features in chunks of 64

struct ChainP {
    const uint16_t* h_bf;            // [M,256] block input: q_in half of the o-projection AND residual of res-norm 1
    const uint16_t* cx;              // [M,256] attention context
    const unsigned char* wpack;      // ttsmi_dense_chain_pack
    int M, F, nchunk, nstages, nhalf;
    const float *bo, *ln1_g, *ln1_b, *b1, *b2, *ln2_g, *ln2_b, *bqkv;
    const uint8_t* row_pad;
    uint32_t thr; float inv_keep; uint64_t seed; const int64_t* step_dev; uint32_t site_ln1, site_ln2;
    float eps;
    uint16_t *a_bf, *xhat1; float* rstd1;
    uint16_t* h1; uint32_t* relu_bits; int bits_wide;
    uint16_t *out_bf, *xhat2; float* rstd2; float* out32;
    uint16_t* qkv;                   // the next block's [M,768], or NULL
    int ablate;                      // measurement build only (TTSMI_CHAIN_ABLATE): 1 no multiplies, 2 no DMA, 8 no in-loop stores
    unsigned long long* dbg;         // measurement build only: per (workgroup, wave) phase stamps, ttsmi_dense_chain_debug
};

__device__ __forceinline__ void ch_dma16(const void* gsrc, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_off) : "memory", "m0");
}
__device__ __forceinline__ void ch_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void ch_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned ch_lds_offset(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ uint2 ch_pack4(float a, float b, float c, float d) {
    bf16x4 h;
    h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
    return *reinterpret_cast<uint2*>(&h);
}
__device__ __forceinline__ float ch_bf(const bf16x8& v, int e) { return (float)v[e]; }
// relu of two packed bf16 values: a negative bf16 is a negative int16 (v_pk_max_i16 against 0; -0.0 becomes +0.0)
typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t ch_relu2(uint32_t w) {
    short2v v = __builtin_bit_cast(short2v, w);
    const short2v z = {0, 0};
    v = __builtin_elementwise_max(v, z);
    return __builtin_bit_cast(uint32_t, v);
}
// bit e = bf16 element e of the 8-element group is > 0 (gemm_k256.hip: kw_pos_bits)
__device__ __forceinline__ uint32_t ch_pos_bits(const uint4& v) {
    auto two = [](uint32_t w) { return (((int32_t)(w << 16) > 0) ? 1u : 0u) | (((int32_t)(w & 0xFFFF0000u) > 0) ? 2u : 0u); };
    return two(v.x) | (two(v.y) << 2) | (two(v.z) << 4) | (two(v.w) << 6);
}

#define CH_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// One stage = 32 weight fragments x one MFMA each, in four groups of eight.  The fragments of group g + 1 are requested
// before the multiplies of group g issue (scheduling barriers pin that order): left to itself hipcc reads two fragments
// into the same registers, waits, multiplies twice - with ONE wave per SIMD nothing else hides the LDS round trip, and the
// matrix pipe idles two thirds of the time (first build of this kernel, ISA reading).  mf(g, i, fragment) multiplies.
template <class MF, class DMA>
__device__ __forceinline__ void ch_stage(const unsigned char* Fs, MF&& mf, DMA&& dma) {
    bf16x8 a0[8], a1[8];
#define CH_FRAG(g, i) (*reinterpret_cast<const bf16x8*>(Fs + ((g) * 8 + (i)) * CH_FRAG_BYTES))
    // multiply group g from `cur` while group g + 1 is read into `nxt`: one fragment read behind every multiply
#define CH_GROUP(cur, nxt, g)                                                        \
    __builtin_amdgcn_sched_barrier(0);                                               \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                  \
        mf(g, i, cur[i]);                                                            \
        nxt[i] = CH_FRAG((g) + 1, i);                                                \
    }                                                                                \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                           \
    }                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                               \
    dma(g);
#pragma unroll
    for (int i = 0; i < 8; ++i) a0[i] = CH_FRAG(0, i);
    CH_GROUP(a0, a1, 0)
    CH_GROUP(a1, a0, 1)
    CH_GROUP(a0, a1, 2)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) mf(3, i, a1[i]);
    __builtin_amdgcn_sched_barrier(0);
    dma(3);
#undef CH_GROUP
#undef CH_FRAG
}

// ---- transposing stores -------------------------------------------------------------------------------------------
// producer: the wave's accumulator-layout values of 64 features (two 32-feature tiles, already bf16) -> its scratch slot
__device__ __forceinline__ void ch_slot_write(unsigned char* slot, int l31, int hh, int u, int g, uint2 v) {
    *reinterpret_cast<uint2*>(slot + (l31 * CH_SLOT_LD + u * 32 + 8 * g + 4 * hh) * 2) = v;
}
// consumer: the slot's 32 rows x 128 bytes -> global rows [row0, row0 + 32) at column col0 (16 bytes per lane, 8 rows per
// instruction); optionally the sign bits of what it stores, in the bit-matrix layout of the K = 256 kernels
__device__ __forceinline__ void ch_slot_flush(const unsigned char* slot, uint16_t* dst, long ld, int col0, int row0, int M, int lane,
                                              uint32_t* bits, int bits_wide, int nbchunk) {
    const int r8 = lane >> 3, c8 = (lane & 7) * 8;
    uint32_t bb[4];
    uint4 v[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) v[it] = *reinterpret_cast<const uint4*>(slot + ((r8 + 8 * it) * CH_SLOT_LD + c8) * 2);
    uint16_t* d0 = dst + (long)(row0 + r8) * ld + col0 + c8;
    if (row0 + 32 <= M) {                              // (wave-uniform: every row of the slot exists - no per-row predicate)
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<uint4*>(d0 + (long)(8 * it) * ld) = v[it];
    } else {
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (row0 + r8 + 8 * it < M) *reinterpret_cast<uint4*>(d0 + (long)(8 * it) * ld) = v[it];
    }
    if (bits != nullptr) {
#pragma unroll
        for (int it = 0; it < 4; ++it) bb[it] = ch_pos_bits(v[it]);
    }
    if (bits != nullptr && row0 < M) {
        // (rows of a 64-row tile: R = (row0 & 32) + r8 + 8 it.  tiles past M are never read back: cdiv(M, 64) tiles exist)
        const long tile = row0 >> 6;
        const int half = (row0 >> 5) & 1;
        if (bits_wide) {
            // gemm_k256_wide_kernel: 64-row x 256-column blocks of 256 threads x two words; thread = (R & 7) * 32 + column / 8,
            // bit 8 (R >> 3) + e of the pair; rows 0..31 of the tile are the first word, 32..63 the second
            const int chunk = col0 >> 8, cb = ((col0 & 255) >> 3) + (lane & 7);
            const uint32_t w = bb[0] | (bb[1] << 8) | (bb[2] << 16) | (bb[3] << 24);
            bits[((tile * nbchunk + chunk) * 256 + r8 * 32 + cb) * 2 + half] = w;
        } else {
            // gemm_k256_kernel: 64-row x 128-column blocks of 256 threads x one word; thread = (R & 15) * 16 + column / 8,
            // bit 8 (R >> 4) + e: this lane holds rows r8 + {0, 16} (+ 32 half) of thread A and r8 + 8 + {0, 16} of thread B
            const int chunk = col0 >> 7, cb = ((col0 & 127) >> 3) + (lane & 7);
            uint16_t* b16 = reinterpret_cast<uint16_t*>(bits);
            const long base = (tile * nbchunk + chunk) * 256;
            b16[(base + r8 * 16 + cb) * 2 + half] = (uint16_t)(bb[0] | (bb[2] << 8));
            b16[(base + (r8 + 8) * 16 + cb) * 2 + half] = (uint16_t)(bb[1] | (bb[3] << 8));
        }
    }
}
// fp32 variant of the pair for the stack's last block (its fp32 output is read by the next layer): one 32-feature tile
__device__ __forceinline__ void ch_slot_flush_f32(const unsigned char* slot, float* dst, int col0, int row0, int M, int lane) {
    const int r8 = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = r8 + 8 * it;
        const float4 v = *reinterpret_cast<const float4*>(slot + (r * 36 + c4) * 4);
        if (row0 + r < M) *reinterpret_cast<float4*>(dst + (long)(row0 + r) * CH_D + col0 + c4) = v;
    }
}

// ---- LayerNorm of a wave's 32 rows, in the accumulator layout -----------------------------------------------------
// Z: the product (8 tiles of 32 features); R: the residual as bf16 fragments (fragment 2j + p = registers 8p.. of tile j);
// on return Y holds LN(keep(Z + bias) + R) * rowmask as bf16 fragments (the next product's B operand and the next
// residual); y / x^ / rstd (and the fp32 y when asked for) are stored through the wave's own scratch slot.
// gamma / beta: the workgroup's staged copies in LDS.
template <bool Y32>
__device__ __forceinline__ void ch_layernorm(f32x16 (&Z)[8], const bf16x8 (&R)[16], bf16x8 (&Y)[16], const ChainP& p,
                                             const float* gamma, const float* beta, uint32_t site, int row, int rowc, int row0, bool padded,
                                             unsigned char* slot, int lane, uint16_t* y_bf, uint16_t* xhat, float* rstd_out, float* y32) {
    // (Z already holds product + bias: the accumulators were initialised with the bias vector)
    const int l31 = lane & 31, hh = lane >> 5;
    const uint64_t key = p.thr ? ttsmi_drop_key(p.seed, p.step_dev, site) : 0;
    const uint32_t rb = ttsmi_row_base(key, (uint32_t)rowc);
    float sum4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = 32 * j + 8 * g + 4 * hh;
            float v[4] = {Z[j][4 * g + 0], Z[j][4 * g + 1], Z[j][4 * g + 2], Z[j][4 * g + 3]};
            if (p.thr) {
                const uint32_t h0 = ttsmi_pair_hash(rb, (uint32_t)c0), h1 = ttsmi_pair_hash(rb, (uint32_t)(c0 + 2));
                v[0] *= ((h0 & 0xFFFFu) >= p.thr) ? p.inv_keep : 0.f;
                v[1] *= ((h0 >> 16) >= p.thr) ? p.inv_keep : 0.f;
                v[2] *= ((h1 & 0xFFFFu) >= p.thr) ? p.inv_keep : 0.f;
                v[3] *= ((h1 >> 16) >= p.thr) ? p.inv_keep : 0.f;
            }
            const bf16x8& rr = R[2 * j + (g >> 1)];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] += ch_bf(rr, 4 * (g & 1) + e);
                Z[j][4 * g + e] = v[e];
                sum4[e] += v[e];
            }
        }
    const float invC = 1.0f / (float)CH_D;
    const float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
    const float mean = (sum + __shfl_xor(sum, 32, 64)) * invC;
    float q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = Z[j][r] - mean;
            q4[r & 3] += v * v;
        }
    const float q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
    const float rstd = __builtin_amdgcn_rsqf((q + __shfl_xor(q, 32, 64)) * invC + p.eps);
    const float nmr = -mean * rstd;                              // x^ = v * rstd - mean * rstd
    if (hh == 0 && row < p.M) rstd_out[row] = rstd;
    // normalise and leave, 64 features (two tiles) per round through the wave's scratch slot
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        uint2 xh_q[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = 2 * cc + u;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * j + 8 * g + 4 * hh;
                const float4 gm = *reinterpret_cast<const float4*>(gamma + c0);
                const float4 bt = *reinterpret_cast<const float4*>(beta + c0);
                const float xh[4] = {fmaf(Z[j][4 * g + 0], rstd, nmr), fmaf(Z[j][4 * g + 1], rstd, nmr), fmaf(Z[j][4 * g + 2], rstd, nmr),
                                     fmaf(Z[j][4 * g + 3], rstd, nmr)};
                float y[4] = {fmaf(xh[0], gm.x, bt.x), fmaf(xh[1], gm.y, bt.y), fmaf(xh[2], gm.z, bt.z), fmaf(xh[3], gm.w, bt.w)};
                if (padded) { y[0] = 0.f; y[1] = 0.f; y[2] = 0.f; y[3] = 0.f; }
                xh_q[u][g] = ch_pack4(xh[0], xh[1], xh[2], xh[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    Y[2 * j + (g >> 1)][4 * (g & 1) + e] = (__bf16)y[e];
                    if constexpr (Y32) Z[j][4 * g + e] = y[e];    // (kept for the fp32 store below; res-norm 1 ignores it)
                }
            }
        }
        // x^
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) ch_slot_write(slot, l31, hh, u, g, xh_q[u][g]);
        ch_lds_fence();
        ch_slot_flush(slot, xhat, CH_D, 64 * cc, row0, p.M, lane, nullptr, 0, 0);
        ch_lds_fence();
        // y (bf16)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = 2 * cc + u;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bf16x8& yy = Y[2 * j + (g >> 1)];
                bf16x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = yy[4 * (g & 1) + e];
                ch_slot_write(slot, l31, hh, u, g, *reinterpret_cast<uint2*>(&h));
            }
        }
        ch_lds_fence();
        ch_slot_flush(slot, y_bf, CH_D, 64 * cc, row0, p.M, lane, nullptr, 0, 0);
        ch_lds_fence();
        if (Y32 && y32 != nullptr) {                               // (template: the stack's last block; run time: its res-norm 2)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = 2 * cc + u;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(slot + (l31 * 36 + 8 * g + 4 * hh) * 4) =
                        make_float4(Z[j][4 * g + 0], Z[j][4 * g + 1], Z[j][4 * g + 2], Z[j][4 * g + 3]);
                ch_lds_fence();
                ch_slot_flush_f32(slot, y32, 32 * j, row0, p.M, lane);
                ch_lds_fence();
            }
        }
    }
}

// accumulators of a full-row product start at its bias vector (staged in LDS)
__device__ __forceinline__ void ch_bias_init(f32x16 (&Z)[8], const float* bias, int hh) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias + 32 * j + 8 * g + 4 * hh);
            Z[j][4 * g + 0] = b4.x; Z[j][4 * g + 1] = b4.y; Z[j][4 * g + 2] = b4.z; Z[j][4 * g + 3] = b4.w;
        }
}

template <bool Y32>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void dense_chain_kernel(ChainP p) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[CH_NRING * CH_STAGE_BYTES + CH_SCR_BYTES + CH_PAR_FLOATS * 4];
    unsigned char* scr = smem + CH_NRING * CH_STAGE_BYTES;
    float* par = reinterpret_cast<float*>(scr + CH_SCR_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * CH_ROWS;
    const int row0 = m0 + wave * 32, row = row0 + l31, rowc = min(row, p.M - 1);
    const int nst = p.nstages;
    const unsigned ring_off = ch_lds_offset(smem);
#ifdef TTSMI_ABLATION_BUILD
    unsigned long long tph[8], twait = 0;
    int nph = 0;
#define CH_STAMP() tph[nph++] = __builtin_readcyclecounter()
#else
#define CH_STAMP()
#endif
    CH_STAMP();
    const int abl = TTSMI_ABLATE_BITS(p.ablate);

    // two of this wave's eight pieces of stage s (pieces [8 wave + 2 g, + 2)) as one m0 set-up and two instructions (the
    // instruction offset applies to the global AND the LDS address); no-op past the end of the stream
    const unsigned char* wsrc = p.wpack + (size_t)wave * CH_NDMA * CH_FRAG_BYTES + lane * 16;
    const unsigned wdst = ring_off + (unsigned)wave * CH_NDMA * CH_FRAG_BYTES;
    auto issue2 = [&](int s, int g) {
        if (s >= nst || (abl & 2)) return;
        const unsigned char* src = wsrc + (size_t)s * CH_STAGE_BYTES + (g >> 1) * 4096;
        const unsigned dst = wdst + (unsigned)(s % CH_NRING) * CH_STAGE_BYTES + (g >> 1) * 4096;
        if (g & 1)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                         ::"v"(src), "s"(dst) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024"
                         ::"v"(src), "s"(dst) : "memory", "m0");
    };
    // Stage s has landed once at most the pieces of the stages BEHIND it are outstanding on every wave: 16 (stages s + 1 and
    // s + 2 were issued during stages s - 2 and s - 1), 8 or 0 near the end of the stream (see the header for why only
    // DMA pieces are counted).  The barrier also retires stage s - 1's slot, which stage s + 3 is then issued into.
    auto stage_begin = [&](int s) -> const unsigned char* {
#ifdef TTSMI_ABLATION_BUILD
        const unsigned long long tw0 = __builtin_readcyclecounter();
#endif
        if (s + 2 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CH_NDMA) : "memory");
        else if (s + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CH_NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ch_barrier();
#ifdef TTSMI_ABLATION_BUILD
        twait += __builtin_readcyclecounter() - tw0;
#endif
        return smem + (s % CH_NRING) * CH_STAGE_BYTES + lane * 16;
    };

    // ---- the wave's 32 rows of [h | ctx] as B fragments in the accumulator's k order (k-group q: features 16q + 4hh +
    // {0..3} and 16q + 8 + 4hh + {0..3}).  Read with full-line accesses (a wave instruction = 4 rows x 256 bytes) and turned
    // into fragments through LDS: the wave's quarter of ring slot 3, which its own DMA pieces only reach during stage 0
    // (8-byte reads straight from the rows were 64 instructions of 32 scattered segments each: 20 k cycles of prologue).
    // Chunk c (16 bytes) of row r sits at position c ^ (r & 15): conflict-free 8-byte fragment reads.
    bf16x8 X[32];
    {
        unsigned char* xs = smem + (CH_NRING - 1) * CH_STAGE_BYTES + wave * 8192;
        const int lr = lane >> 4, lc = lane & 15;
        // (four named arrays, not raw[4][8]: hipcc 7.2 sent the two-dimensional array to scratch)
        uint4 raw0[8], raw1[8], raw2[8], raw3[8];
#define CH_XLOAD(dst, base, half)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                    \
        const int r = row0 + 4 * i + lr;              /* (a select, not a clamped index: rowgemm.hip's note on hipcc 7.2) */ \
        dst[i] = r < p.M ? *reinterpret_cast<const uint4*>((base) + (long)r * CH_D + (half) * 128 + lc * 8)             \
                         : make_uint4(0u, 0u, 0u, 0u);                                                                  \
    }
        CH_XLOAD(raw0, p.h_bf, 0)
        CH_XLOAD(raw1, p.h_bf, 1)
        CH_XLOAD(raw2, p.cx, 0)
        CH_XLOAD(raw3, p.cx, 1)
#undef CH_XLOAD
#define CH_XFRAGS(src, t)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                    \
        const int r = 4 * i + lr;                                                                                      \
        *reinterpret_cast<uint4*>(xs + r * 256 + ((lc ^ (r & 15)) << 4)) = src[i];                                     \
    }                                                                                                                  \
    ch_lds_fence();                                                                                                    \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                                    \
        const uint2 lo = *reinterpret_cast<const uint2*>(xs + l31 * 256 + (((2 * q) ^ (l31 & 15)) << 4) + 8 * hh);     \
        const uint2 hi = *reinterpret_cast<const uint2*>(xs + l31 * 256 + (((2 * q + 1) ^ (l31 & 15)) << 4) + 8 * hh); \
        const uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);                                                            \
        X[8 * (t) + q] = *reinterpret_cast<const bf16x8*>(&v);                                                         \
    }                                                                                                                  \
    ch_lds_fence();
        CH_XFRAGS(raw0, 0)
        CH_XFRAGS(raw1, 1)
        CH_XFRAGS(raw2, 2)
        CH_XFRAGS(raw3, 3)
#undef CH_XFRAGS
    }
    const bool padded = p.row_pad != nullptr && p.row_pad[rowc] != 0;
    // ---- parameter vectors -> LDS (13 KB; published by the first stage's barrier)
    {
        auto stage_vec = [&](const float* src, int off, int n) {
            for (int i = tid * 4; i < n; i += 256 * 4) *reinterpret_cast<float4*>(par + off + i) = *reinterpret_cast<const float4*>(src + i);
        };
        stage_vec(p.bo, CH_P_BO, CH_D); stage_vec(p.ln1_g, CH_P_G1, CH_D); stage_vec(p.ln1_b, CH_P_BE1, CH_D);
        stage_vec(p.b2, CH_P_B2, CH_D); stage_vec(p.ln2_g, CH_P_G2, CH_D); stage_vec(p.ln2_b, CH_P_BE2, CH_D);
        stage_vec(p.b1, CH_P_B1, p.F);
        if (p.qkv != nullptr) stage_vec(p.bqkv, CH_P_BQ, 3 * CH_D);
    }
#pragma unroll
    for (int s = 0; s < CH_NRING - 1; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g) issue2(s, g);

    f32x16 Z[8];
    CH_STAMP();

    // ---- two halves, ONE copy of the LayerNorm code (p.nhalf is 2 at run time: the compiler cannot unroll the loop):
    //   half 0: o-projection, 8 stages of (4 k-groups x 8 output tiles)             -> res-norm 1
    //   half 1: FFN, per 64 hidden features a stage of a . W1 and one of h1 . W2    -> res-norm 2
    // Y (the LayerNorm's output fragments = the next products' B operand = the next residual) lives in the registers of the h
    // half of X: res-norm 1 reads its residual there and overwrites it in place, as res-norm 2 does with a.
    int S = 0;
    unsigned char* slot = scr + wave * CH_SLOT_BYTES;              // the wave's own transposing scratch
    bf16x8(&Y)[16] = *reinterpret_cast<bf16x8(*)[16]>(&X[0]);
    const int nbchunk = p.bits_wide ? p.F / 256 : p.F / 128;
    for (int half = 0; half < p.nhalf; ++half) {
        if (half == 0) {
#pragma unroll
            for (int s = 0; s < CH_WO_STAGES; ++s) {
                const unsigned char* Fs = stage_begin(S);
                if (s == 0) ch_bias_init(Z, par + CH_P_BO, hh);     // (the staged vectors are published by the first barrier)
                ch_stage(Fs, [&](int kq, int j, const bf16x8& a) { if (!(abl & 1)) Z[j] = CH_MFMA(a, X[4 * s + kq], Z[j]); },
                         [&](int g) { issue2(S + CH_NRING - 1, g); });
                ++S;
            }
        } else {
            ch_bias_init(Z, par + CH_P_B2, hh);
            for (int c = 0; c < p.nchunk; ++c) {
                const unsigned char* Fs = stage_begin(S);
                f32x16 H[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 b4 = *reinterpret_cast<const float4*>(par + CH_P_B1 + 64 * c + 32 * u + 8 * g + 4 * hh);
                        H[u][4 * g + 0] = b4.x; H[u][4 * g + 1] = b4.y; H[u][4 * g + 2] = b4.z; H[u][4 * g + 3] = b4.w;
                    }
                ch_stage(Fs, [&](int q4, int i, const bf16x8& a) { if (!(abl & 1)) H[i & 1] = CH_MFMA(a, Y[q4 * 4 + (i >> 1)], H[i & 1]); },
                         [&](int g) { issue2(S + CH_NRING - 1, g); });
                ++S;
                bf16x8 hf[4];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint2 h = ch_pack4(H[u][4 * g + 0], H[u][4 * g + 1], H[u][4 * g + 2], H[u][4 * g + 3]);
                        h.x = ch_relu2(h.x);
                        h.y = ch_relu2(h.y);
                        const bf16x4 hb = *reinterpret_cast<const bf16x4*>(&h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) hf[2 * u + (g >> 1)][4 * (g & 1) + e] = hb[e];
                        ch_slot_write(slot, l31, hh, u, g, h);
                    }
                Fs = stage_begin(S);
                // the wave's h1 chunk leaves now, as early in this stage as possible: these four stores are the oldest thing
                // on vmcnt by the next stage's counted wait
                if (!(abl & 8)) ch_slot_flush(slot, p.h1, p.F, 64 * c, row0, p.M, lane, p.relu_bits, p.bits_wide, nbchunk);
                ch_stage(Fs, [&](int g4, int j, const bf16x8& a) { if (!(abl & 1)) Z[j] = CH_MFMA(a, hf[g4], Z[j]); },
                         [&](int g) { issue2(S + CH_NRING - 1, g); });
                ++S;
            }
            ch_lds_fence();
        }
        CH_STAMP();
        ch_layernorm<Y32>(Z, Y, Y, p, par + (half ? CH_P_G2 : CH_P_G1), par + (half ? CH_P_BE2 : CH_P_BE1), half ? p.site_ln2 : p.site_ln1, row,
                          rowc, row0, padded, slot, lane, half ? p.out_bf : p.a_bf, half ? p.xhat2 : p.xhat1, half ? p.rstd2 : p.rstd1,
                          half ? p.out32 : nullptr);
        CH_STAMP();
    }

    // ---- the next block's qkv projection: 12 stages of (2 output tiles x 16 k-groups)
    if (p.qkv != nullptr) {
        for (int s = 0; s < CH_QKV_STAGES; ++s) {
            const unsigned char* Fs = stage_begin(S);
            if (s > 0 && !(abl & 8)) ch_slot_flush(slot, p.qkv, 3 * CH_D, 64 * (s - 1), row0, p.M, lane, nullptr, 0, 0);
            f32x16 acc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b4 = *reinterpret_cast<const float4*>(par + CH_P_BQ + 64 * s + 32 * u + 8 * g + 4 * hh);
                    acc[u][4 * g + 0] = b4.x; acc[u][4 * g + 1] = b4.y; acc[u][4 * g + 2] = b4.z; acc[u][4 * g + 3] = b4.w;
                }
            ch_stage(Fs, [&](int q4, int i, const bf16x8& a) { if (!(abl & 1)) acc[i & 1] = CH_MFMA(a, Y[q4 * 4 + (i >> 1)], acc[i & 1]); },
                     [&](int g) { issue2(S + CH_NRING - 1, g); });
            ++S;
            ch_lds_fence();                                        // (the flush above has read the slot)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    ch_slot_write(slot, l31, hh, u, g, ch_pack4(acc[u][4 * g + 0], acc[u][4 * g + 1], acc[u][4 * g + 2], acc[u][4 * g + 3]));
        }
        ch_lds_fence();
        ch_slot_flush(slot, p.qkv, 3 * CH_D, 64 * (CH_QKV_STAGES - 1), row0, p.M, lane, nullptr, 0, 0);
    }
#ifdef TTSMI_ABLATION_BUILD
    CH_STAMP();
    if (p.dbg && lane == 0) {              // [workgroup][wave][8]: start, then the six phase durations, then the time in stage waits
        unsigned long long* o = p.dbg + ((long)blockIdx.x * CH_NW + wave) * 8;
        o[0] = tph[0];
        for (int i = 1; i < 7; ++i) o[i] = tph[i] - tph[i - 1];      // prologue, o-projection, LN1, FFN, LN2, qkv
        o[7] = twait;
    }
#endif
}

// ---- the weight stream ----------------------------------------------------------------------------------------------
struct ChainPackP {
    const uint16_t *wo_t, *w1_t, *w2_t, *wqkv_t;      // [256][512], [F][256], [256][F], [768][256] (W^T as stored by the shadow set)
    uint16_t* out;
    int F, nchunk, nstages;
    int wo_stages;                                    // 8 (forward stream) or 0 (backward stream: no product in front of the FFN pair)
};
__global__ __launch_bounds__(256) void dense_chain_pack_kernel(ChainPackP p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // one 16-byte lane item of one fragment
    const long total = (long)p.nstages * CH_STAGE_FRAGS * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63), f = (int)((idx >> 6) & 31), S = (int)(idx >> 11);
    const int l31 = lane & 31, hh = lane >> 5;
    const uint16_t* src;
    long ld;
    int n, kbase;
    if (S < CH_WO_STAGES) {
        const int kq = f >> 3, j = f & 7;
        src = p.wo_t; ld = 2 * CH_D; n = 32 * j + l31; kbase = 16 * (4 * S + kq);
    } else if (S < CH_WO_STAGES + 2 * p.nchunk) {
        const int t = S - CH_WO_STAGES, c = t >> 1;
        if ((t & 1) == 0) {
            const int q = f >> 1, u = f & 1;
            src = p.w1_t; ld = CH_D; n = 64 * c + 32 * u + l31; kbase = 16 * q;
        } else {
            const int g4 = f >> 3, j = f & 7;
            src = p.w2_t; ld = p.F; n = 32 * j + l31; kbase = 64 * c + 16 * g4;
        }
    } else {
        const int s = S - CH_WO_STAGES - 2 * p.nchunk, q = f >> 1, u = f & 1;
        src = p.wqkv_t; ld = CH_D; n = 64 * s + 32 * u + l31; kbase = 16 * q;
    }
    const uint16_t* r = src + (long)n * ld + kbase + 4 * hh;
    const uint2 lo = *reinterpret_cast<const uint2*>(r), hi = *reinterpret_cast<const uint2*>(r + 8);
    *reinterpret_cast<uint4*>(p.out + idx * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
}

#include "chain16.h"
#include "chain16b.h"

// which form of the kernel (and of its weight stream) this process uses: TTSMI_DENSE_CHAIN_FORM = 16 (eight 16-row waves on
// 16x16x32 MFMAs, two waves per SIMD) or 32 (four 32-row waves on 32x32x16, one wave per SIMD); read once
static int chain_form() {
    TTSMI_KNOB(form, "TTSMI_DENSE_CHAIN_FORM", 16);
    return form == 32 ? 32 : 16;
}

static int chain_stages(int F, int with_qkv) { return CH_WO_STAGES + 2 * (F / 64) + (with_qkv ? CH_QKV_STAGES : 0); }

#ifdef TTSMI_ABLATION_BUILD
static unsigned long long* g_chain_dbg = nullptr;
extern "C" int ttsmi_dense_chain_debug(void* buf) { g_chain_dbg = (unsigned long long*)buf; return 0; }     // [workgroups][4][8] uint64
#endif

extern "C" {

size_t ttsmi_dense_chain_pack_bytes(int F, int with_qkv) {
    return F > 0 && F % 64 == 0 ? (size_t)chain_stages(F, with_qkv) * CH_STAGE_BYTES : 0;
}

int ttsmi_dense_chain_pack(const uint16_t* wo_t, const uint16_t* w1_t, const uint16_t* w2_t, const uint16_t* wqkv_next_t, int F,
                           void* out, size_t out_bytes, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(wo_t && w1_t && w2_t && out, "dense_chain_pack: null pointer");
    TTSMI_CHECK_ARG(F > 0 && F % 64 == 0, "dense_chain_pack: F must be a multiple of 64 (got %d)", F);
    TTSMI_CHECK_ARG(out_bytes >= ttsmi_dense_chain_pack_bytes(F, wqkv_next_t != nullptr), "dense_chain_pack: output buffer too small");
    TTSMI_CHECK_ARG(((((uintptr_t)wo_t) | ((uintptr_t)w1_t) | ((uintptr_t)w2_t) | ((uintptr_t)wqkv_next_t)) & 7) == 0 && (((uintptr_t)out) & 15) == 0,
                    "dense_chain_pack: operands must be 8-byte (output: 16-byte) aligned");
    ChainPackP p;
    p.wo_t = wo_t; p.w1_t = w1_t; p.w2_t = w2_t; p.wqkv_t = wqkv_next_t; p.out = (uint16_t*)out;
    p.F = F; p.nchunk = F / 64; p.nstages = chain_stages(F, wqkv_next_t != nullptr);
    p.wo_stages = CH_WO_STAGES;
    const long total = (long)p.nstages * CH_STAGE_FRAGS * 64;
    if (chain_form() == 16) hipLaunchKernelGGL(dense_chain16_pack_kernel, dim3(ttsmi_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(dense_chain_pack_kernel, dim3(ttsmi_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
    TTSMI_CHECK_LAUNCH("dense_chain_pack");
    return TTSMI_OK;
}

int ttsmi_dense_chain_supported(int M, int d, int F) { return M > 0 && d == CH_D && F >= 64 && F % 64 == 0 && F <= CH_MAXF; }

int ttsmi_dense_chain_fwd(const uint16_t* h_bf, const uint16_t* ctx, const void* wpack, size_t wpack_bytes, int M, int F,
                          const float* bo, const float* ln1_g, const float* ln1_b, const float* b1, const float* b2,
                          const float* ln2_g, const float* ln2_b, const float* bqkv_next, const uint8_t* row_pad, float p_drop,
                          uint64_t seed, const int64_t* step_dev, uint32_t site_ln1, uint32_t site_ln2, float eps, uint16_t* a_bf,
                          uint16_t* xhat1, float* rstd1, uint16_t* h1, void* relu_bits, int relu_bits_layout, uint16_t* out_bf,
                          uint16_t* xhat2, float* rstd2, float* out32, uint16_t* qkv_next, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(h_bf && ctx && wpack && bo && ln1_g && ln1_b && b1 && b2 && ln2_g && ln2_b && a_bf && xhat1 && rstd1 && h1 &&
                        out_bf && xhat2 && rstd2,
                    "dense_chain_fwd: null pointer");
    TTSMI_CHECK_ARG(ttsmi_dense_chain_supported(M, CH_D, F), "dense_chain_fwd: unsupported shape M=%d F=%d", M, F);
    TTSMI_CHECK_ARG((qkv_next == nullptr) == (bqkv_next == nullptr), "dense_chain_fwd: qkv_next and bqkv_next go together");
    TTSMI_CHECK_ARG(wpack_bytes >= ttsmi_dense_chain_pack_bytes(F, qkv_next != nullptr), "dense_chain_fwd: weight stream too short");
    TTSMI_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dense_chain_fwd: bad dropout rate");
    TTSMI_CHECK_ARG(((((uintptr_t)h_bf) | ((uintptr_t)ctx) | ((uintptr_t)wpack) | ((uintptr_t)a_bf) | ((uintptr_t)xhat1) | ((uintptr_t)h1) |
                      ((uintptr_t)out_bf) | ((uintptr_t)xhat2) | ((uintptr_t)out32) | ((uintptr_t)qkv_next) | ((uintptr_t)bo) |
                      ((uintptr_t)ln1_g) | ((uintptr_t)ln1_b) | ((uintptr_t)b1) | ((uintptr_t)b2) | ((uintptr_t)ln2_g) | ((uintptr_t)ln2_b) |
                      ((uintptr_t)bqkv_next)) & 15) == 0 && (((uintptr_t)relu_bits) & 7) == 0,
                    "dense_chain_fwd: operands must be 16-byte aligned");
    ChainP p;
    memset(&p, 0, sizeof(p));
    p.h_bf = h_bf; p.cx = ctx; p.wpack = (const unsigned char*)wpack; p.M = M; p.F = F; p.nchunk = F / 64;
    p.nstages = chain_stages(F, qkv_next != nullptr);
    p.nhalf = 2;
    p.bo = bo; p.ln1_g = ln1_g; p.ln1_b = ln1_b; p.b1 = b1; p.b2 = b2; p.ln2_g = ln2_g; p.ln2_b = ln2_b; p.bqkv = bqkv_next;
    p.row_pad = row_pad;
    p.thr = p_drop > 0.f ? ttsmi_drop_threshold(p_drop) : 0u;
    p.inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    p.seed = seed; p.step_dev = step_dev; p.site_ln1 = site_ln1; p.site_ln2 = site_ln2; p.eps = eps;
    p.a_bf = a_bf; p.xhat1 = xhat1; p.rstd1 = rstd1; p.h1 = h1; p.relu_bits = (uint32_t*)relu_bits;
    // the bit matrix is read back by ttsmi_hgemm_k256_masked_bits: its 256-column variant from 16 384 rows (gemm_k256.hip: kw_launch)
    TTSMI_KNOB(wide, "TTSMI_HGEMM_K256_WIDE", 1);
    p.bits_wide = (wide && F % 256 == 0 && (M >= 16384 || wide > 1)) ? 1 : 0;
    if (relu_bits != nullptr && relu_bits_layout == 1) {
        TTSMI_CHECK_ARG(chain_form() == 16, "dense_chain_fwd: the backward chain's bit layout needs the 16-row form");
        p.bits_wide = 2;
    }
    if (relu_bits != nullptr && !p.bits_wide) TTSMI_CHECK_ARG(F % 128 == 0, "dense_chain_fwd: the bit matrix needs F %% 128 == 0");
    p.out_bf = out_bf; p.xhat2 = xhat2; p.rstd2 = rstd2; p.out32 = out32; p.qkv = qkv_next;
#ifdef TTSMI_ABLATION_BUILD
    TTSMI_ABLATE_KNOB(ablate, "TTSMI_CHAIN_ABLATE");
    p.ablate = ablate;
    p.dbg = g_chain_dbg;
#endif
    if (chain_form() == 16) {
        ttsmi_note_kernel("dense_chain16_kernel");
        if (out32 != nullptr) TTSMI_LAUNCH_EV(dense_chain16_kernel<true>, dim3(ttsmi_cdiv(M, C16_ROWS)), dim3(C16_NW * 64), 0, (hipStream_t)stream, p);
        else TTSMI_LAUNCH_EV(dense_chain16_kernel<false>, dim3(ttsmi_cdiv(M, C16_ROWS)), dim3(C16_NW * 64), 0, (hipStream_t)stream, p);
        TTSMI_CHECK_LAUNCH("dense_chain_fwd");
        return TTSMI_OK;
    }
    ttsmi_note_kernel("dense_chain_kernel");
    if (out32 != nullptr) TTSMI_LAUNCH_EV(dense_chain_kernel<true>, dim3(ttsmi_cdiv(M, CH_ROWS)), dim3(CH_NW * 64), 0, (hipStream_t)stream, p);
    else TTSMI_LAUNCH_EV(dense_chain_kernel<false>, dim3(ttsmi_cdiv(M, CH_ROWS)), dim3(CH_NW * 64), 0, (hipStream_t)stream, p);
    TTSMI_CHECK_LAUNCH("dense_chain_fwd");
    return TTSMI_OK;
}

// ---- backward chain (chain16b.h) -----------------------------------------------------------------------------------
static int chain_bwd_stages(int F) { return 2 * (F / 64) + C16B_CTX_STAGES; }

size_t ttsmi_dense_chain_bwd_pack_bytes(int F) { return F > 0 && F % 64 == 0 ? (size_t)chain_bwd_stages(F) * CH_STAGE_BYTES : 0; }

int ttsmi_dense_chain_bwd_supported(int M, int d, int F) { return chain_form() == 16 && ttsmi_dense_chain_supported(M, d, F) && M >= 1; }

/* rows of dgamma / dbeta partials ttsmi_dense_chain_bwd leaves in part_ws: one per 128-row workgroup */
int ttsmi_dense_chain_bwd_nparts(int M) { return ttsmi_cdiv(M, C16_ROWS); }

/* w1_b [256][F], w2_b [F][256], wo_b [512][256]: the weights AS STORED (bf16 shadows) */
int ttsmi_dense_chain_bwd_pack(const uint16_t* w1_b, const uint16_t* w2_b, const uint16_t* wo_b, int F, void* out, size_t out_bytes,
                               ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(w1_b && w2_b && wo_b && out, "dense_chain_bwd_pack: null pointer");
    TTSMI_CHECK_ARG(F > 0 && F % 64 == 0, "dense_chain_bwd_pack: F must be a multiple of 64 (got %d)", F);
    TTSMI_CHECK_ARG(out_bytes >= ttsmi_dense_chain_bwd_pack_bytes(F), "dense_chain_bwd_pack: output buffer too small");
    TTSMI_CHECK_ARG(((((uintptr_t)w1_b) | ((uintptr_t)w2_b) | ((uintptr_t)wo_b)) & 7) == 0 && (((uintptr_t)out) & 15) == 0,
                    "dense_chain_bwd_pack: operands must be 8-byte (output: 16-byte) aligned");
    // the forward packer's three roles with the backward's matrices: "W1^T [F][256]" = W2 as stored (rows = hidden feature, k = d),
    // "W2^T [256][F]" = W1 as stored (rows = d, k = hidden feature), the tail = the ctx half of Wo as stored ([256][256]: 4 stages)
    ChainPackP p;
    p.wo_t = nullptr; p.w1_t = w2_b; p.w2_t = w1_b; p.wqkv_t = wo_b + (long)CH_D * CH_D; p.out = (uint16_t*)out;
    p.F = F; p.nchunk = F / 64; p.nstages = chain_bwd_stages(F); p.wo_stages = 0;
    const long total = (long)p.nstages * CH_STAGE_FRAGS * 64;
    hipLaunchKernelGGL(dense_chain16_pack_kernel, dim3(ttsmi_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
    TTSMI_CHECK_LAUNCH("dense_chain_bwd_pack");
    return TTSMI_OK;
}

int ttsmi_dense_chain_bwd(const uint16_t* df, const uint16_t* da, const uint16_t* xhat1, const float* rstd1, const float* ln1_g,
                          const uint8_t* row_pad, const void* relu_bits_lane, const void* wpack, size_t wpack_bytes, int M, int F,
                          float p_drop, uint64_t seed, const int64_t* step_dev, uint32_t site_ln1, uint16_t* dh1, uint16_t* d_o,
                          void* dres, int dres_is_bf16, uint16_t* dctx, void* part_ws, size_t part_ws_bytes, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(df && da && xhat1 && rstd1 && ln1_g && relu_bits_lane && wpack && dh1 && d_o && dres && dctx && part_ws,
                    "dense_chain_bwd: null pointer");
    TTSMI_CHECK_ARG(ttsmi_dense_chain_bwd_supported(M, CH_D, F), "dense_chain_bwd: unsupported shape M=%d F=%d (or TTSMI_DENSE_CHAIN_FORM=32)", M, F);
    TTSMI_CHECK_ARG(wpack_bytes >= ttsmi_dense_chain_bwd_pack_bytes(F), "dense_chain_bwd: weight stream too short");
    TTSMI_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "dense_chain_bwd: bad dropout rate");
    const int nparts = ttsmi_dense_chain_bwd_nparts(M);
    TTSMI_CHECK_ARG(part_ws_bytes >= ttsmi_layernorm_partials_bytes(nparts, CH_D), "dense_chain_bwd: partial-sum workspace too small");
    TTSMI_CHECK_ARG(((((uintptr_t)df) | ((uintptr_t)da) | ((uintptr_t)xhat1) | ((uintptr_t)wpack) | ((uintptr_t)dh1) | ((uintptr_t)d_o) |
                      ((uintptr_t)dres) | ((uintptr_t)dctx) | ((uintptr_t)ln1_g) | ((uintptr_t)part_ws)) & 15) == 0 &&
                        (((uintptr_t)relu_bits_lane) & 1) == 0,
                    "dense_chain_bwd: operands must be 16-byte aligned");
    ChainBP p;
    memset(&p, 0, sizeof(p));
    p.df = df; p.da = da; p.xhat = xhat1; p.rstd = rstd1; p.gamma = ln1_g; p.row_pad = row_pad;
    p.bits16 = (const uint16_t*)relu_bits_lane; p.wpack = (const unsigned char*)wpack;
    p.M = M; p.F = F; p.nchunk = F / 64; p.nstages = chain_bwd_stages(F); p.nparts = nparts;
    p.thr = p_drop > 0.f ? ttsmi_drop_threshold(p_drop) : 0u;
    p.inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    p.seed = seed; p.step_dev = step_dev; p.site = site_ln1;
    p.dh1 = dh1; p.d_o = d_o; p.dctx = dctx; p.dres = dres; p.dres_bf16 = dres_is_bf16 ? 1 : 0; p.part = (float*)part_ws;
    ttsmi_note_kernel("dense_chain16_bwd_kernel");
    TTSMI_LAUNCH_EV(dense_chain16_bwd_kernel, dim3(nparts), dim3(C16_NW * 64), 0, (hipStream_t)stream, p);
    TTSMI_CHECK_LAUNCH("dense_chain_bwd");
    return TTSMI_OK;
}

}  // extern "C"
