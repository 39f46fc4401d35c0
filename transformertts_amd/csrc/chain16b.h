// The row-local chain of a dense block's BACKWARD between its two res-norms, as one launch (16-row form; included by chain.hip):
//
//   dh1 = (df . W2^T) * [h1 > 0]                          (d FFN hidden; model/layers.py:99-100 differentiated)
//   g   = da + dh1 . W1^T                                 (upstream gradient of res-norm 1: residual path + FFN path)
//   dz  = rstd1 (t - mean(t) - x^1 mean(t x^1)),  t = g * rowmask * gamma1          (LayerNorm backward, x^ form)
//   d_o = keep(dz),  dh = dz                              (dropout of the o-projection's output; residual gradient)
//   dctx = d_o . Wo[d:2d]^T                               (the ctx half of Dense(concat([q_in, ctx])), layers.py:148-149)
//   + the workgroup's partial row of dgamma1 / dbeta1
//
// It replaces three launches of ttsmi_dense_block_bwd - ttsmi_hgemm_k256_masked_bits, ttsmi_hgemm_ln_bwd_dual_h (K = F) and
// the dctx GEMM - which re-read dh1 (59 MB per decoder block) and d_o from HBM.  Inputs df / da come from the res-norm 2
// backward that runs in the epilogue of the block above (or from ttsmi_layernorm_bwd_xhat_h); x^1, rstd1 and the ReLU bits
// from the forward chain (bits in ITS lane layout: word (row / 16, chunk, lane) holds bit 4 u + r of feature
// 64 chunk + 16 u + 4 (lane >> 4) + r, row lane & 15).  Everything the weight-gradient stream needs is written: dh1, d_o.
// Same structure as dense_chain16_kernel: 36 stages at F = 1 024 (per 64 hidden features: df . W2^T, then dh1 . W1^T; then
// four stages of d_o . Wo_ctx^T), packed by the same kernel from the as-stored (`_b`) bf16 shadows.
#pragma once

struct ChainBP {
    const uint16_t *df, *da;             // [M,256] bf16
    const uint16_t* xhat; const float* rstd; const float* gamma;
    const uint8_t* row_pad;
    const uint16_t* bits16;              // the forward chain's lane-layout ReLU bits
    const unsigned char* wpack;
    int M, F, nchunk, nstages, nparts;
    uint32_t thr; float inv_keep; uint64_t seed; const int64_t* step_dev; uint32_t site;
    uint16_t *dh1, *d_o, *dctx;
    void* dres; int dres_bf16;
    float* part;                         // [2 nparts][256]: dgamma partial rows, then dbeta partial rows
    float* xbuf; uint32_t* xflag;        // SPLIT form: the exchange buffers of the forward's (chain16.h)
};

#define C16B_CTX_STAGES 4

// SPLIT: two workgroups per 64-row tile, as in the forward (chain16.h): each streams half of the hidden chunks (the partial sums
// of g = da + dh1 . W1^T are exchanged before res-norm 1's backward, which both run) and half of the dctx columns; the first of
// the pair stores d_o / dres and the tile's parameter-gradient partial row.
template <int NW, int NL = 0, bool SPLIT = false>
__global__ __launch_bounds__((NW + NL) * 64, 1) __attribute__((amdgpu_waves_per_eu((NW + 3) / 4, (NW + NL + 3) / 4))) void dense_chain16_bwd_kernel(ChainBP p) {
    static_assert(!SPLIT || NL > 0, "the split form streams through loader waves");
    constexpr int C16_ROWS = NW * 16, C16_NDMA = CH_STAGE_FRAGS / NW;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[CH_NRING * CH_STAGE_BYTES + C16_SCR_BYTES + 2 * CH_D * 4];
    unsigned char* scr = smem + CH_NRING * CH_STAGE_BYTES;
    float* gam = reinterpret_cast<float*>(scr + C16_SCR_BYTES);          // gamma1, staged
    const int tid = threadIdx.x, lane = tid & 63, t = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int side = SPLIT ? (blockIdx.x >> 3) & 1 : 0;
    const int tile = SPLIT ? (blockIdx.x >> 4) * 8 + (blockIdx.x & 7) : blockIdx.x;
    const int m0 = tile * C16_ROWS;
    const int row0 = m0 + wave * 16, row = row0 + t, rowc = min(row, p.M - 1);
    const int nc_w = SPLIT ? p.nchunk / 2 : p.nchunk, c_first = side * nc_w;
    const int nq_w = SPLIT ? C16B_CTX_STAGES / 2 : C16B_CTX_STAGES;
    const int nst = SPLIT ? 2 * nc_w + nq_w : p.nstages;
    auto packed_stage = [=](int s) -> int {
        if (!SPLIT) return s;
        if (s < 2 * nc_w) return s + 2 * c_first;
        return s + 2 * (p.nchunk - nc_w) + side * nq_w;
    };
    const unsigned ring_off = ch_lds_offset(smem);
    const unsigned char* wsrc = p.wpack + (size_t)wave * C16_NDMA * CH_FRAG_BYTES + lane * 16;
    const unsigned wdst = ring_off + (unsigned)wave * C16_NDMA * CH_FRAG_BYTES;
    if constexpr (NL > 0) {
        if (wave >= NW) {             // loader waves (chain16.h): the one extra barrier sits behind the first dctx stage's
            c16_loader_loop<NL>(p.wpack, ring_off, nst, wave - NW, lane, 2 * nc_w, packed_stage);
            return;
        }
    }
    auto issue2 = [&](int s, int g) {
        if constexpr (NL == 0) {
        if (s >= nst) return;
        c16_issue_half<C16_NDMA>(wsrc + (size_t)s * CH_STAGE_BYTES, __builtin_amdgcn_readfirstlane(wdst + (unsigned)(s % CH_NRING) * CH_STAGE_BYTES), g & 1);
        }
    };
    auto stage_begin = [&](int s) -> const unsigned char* {
        if constexpr (NL == 0) {
            if (s + 2 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C16_NDMA) : "memory");
            else if (s + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C16_NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ch_barrier();
        return smem + (s % CH_NRING) * CH_STAGE_BYTES + lane * 16;
    };

    // ---- the wave's 16 rows of df, da and x^1 as fragments in the accumulator's feature order (chain16.h: the staging of X)
    bf16x8 DF[8], DA[8], XH[8];
    {
        unsigned char* xs = smem + (CH_NRING - 1) * CH_STAGE_BYTES + wave * 4096;
        const int lr = lane >> 4, lc = lane & 15;
        uint4 raw0[4], raw1[4], raw2[4], raw3[4], raw4[4], raw5[4];
#define C16B_XLOAD(dst, base, half)                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                    \
        const int r = row0 + 4 * i + lr;                                                                               \
        dst[i] = r < p.M ? *reinterpret_cast<const uint4*>((base) + (long)r * CH_D + (half) * 128 + lc * 8)             \
                         : make_uint4(0u, 0u, 0u, 0u);                                                                  \
    }
        C16B_XLOAD(raw0, p.df, 0)
        C16B_XLOAD(raw1, p.df, 1)
        C16B_XLOAD(raw2, p.da, 0)
        C16B_XLOAD(raw3, p.da, 1)
        C16B_XLOAD(raw4, p.xhat, 0)
        C16B_XLOAD(raw5, p.xhat, 1)
#undef C16B_XLOAD
#define C16B_XFRAGS(src, dstarr, base)                                                                                 \
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
        C16B_XFRAGS(raw0, DF, 0)
        C16B_XFRAGS(raw1, DF, 4)
        C16B_XFRAGS(raw2, DA, 0)
        C16B_XFRAGS(raw3, DA, 4)
        C16B_XFRAGS(raw4, XH, 0)
        C16B_XFRAGS(raw5, XH, 4)
#undef C16B_XFRAGS
    }
    const bool padded = p.row_pad != nullptr && p.row_pad[rowc] != 0;
    float rstd = p.rstd[rowc];
    // (everything that is loaded is waited for HERE, in front of the first DMA issue: a compiler-placed wait at a later first use
    // would be vmcnt(0) and drain the ring's prefetch - the compiler does not count the inline-asm DMA instructions)
    const uint64_t key = p.thr ? ttsmi_drop_key(p.seed, p.step_dev, p.site) : 0;
    asm volatile("" : "+v"(rstd));
    for (int i = tid * 4; i < CH_D; i += NW * 64 * 4) *reinterpret_cast<float4*>(gam + i) = *reinterpret_cast<const float4*>(p.gamma + i);
#pragma unroll
    for (int s = 0; s < CH_NRING - 1; ++s) {
        issue2(s, 0);
        issue2(s, 1);
    }

    // the accumulators of g start at the residual path's gradient da
    f32x4v Z[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) Z[j][e] = side == 0 ? ch_bf(DA[j >> 1], 4 * (j & 1) + e) : 0.f;

    int S = 0;
    unsigned char* slot = scr + wave * C16_SLOT_BYTES;
    const long tile16 = min(row0, p.M - 1) >> 4;         // the wave's 16-row tile (the bit words are laid out per tile; clamped past M)
    static_assert(C16B_CTX_STAGES >= CH_NRING - 1, "the chunk loop's counted wait for the ReLU word assumes a DMA issue in every one of its stages");
    const uint16_t* bitp = p.bits16 + (tile16 * p.nchunk + c_first) * 64 + lane;
    for (int cc_ = 0; cc_ < nc_w; ++cc_) {
        const int c = c_first + cc_;
        // The chunk's ReLU word.  A compiler-tracked load here costs a full drain: the compiler does not see the LDS-DMA
        // instructions (inline asm), so its wait in front of the first use is vmcnt(0) - behind the stage that has just
        // put four more DMA pieces in flight.  Issued by hand in front of the stage's barrier and waited for with the count
        // of what is younger than it (exactly this stage's four DMA pieces; the chunk loop never runs out of stages to
        // fetch, see the assert), it costs nothing.
        uint32_t bits;
        asm volatile("global_load_ushort %0, %1, off" : "=v"(bits) : "v"(bitp) : "memory");
        bitp += 64;
        const unsigned char* Fs = stage_begin(S);
        f32x4v H[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) H[u][e] = 0.f;
        c16_stage(Fs, [&](int g, int i, const bf16x8& a) { H[i & 3] = C16_MFMA(a, DF[2 * g + (i >> 2)], H[i & 3]); },
                  [&](int g) { issue2(S + CH_NRING - 1, g); });
        ++S;
        // (with loader waves nothing of this wave is younger than the word's load: a plain drain - its stores are a stage old)
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(bits) : "n"(NL > 0 ? 0 : C16_NDMA) : "memory");
        bf16x8 hf[2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((bits >> (4 * u + e)) & 1u) ? H[u][e] : 0.f;
            const uint2 h = ch_pack4(v[0], v[1], v[2], v[3]);
            const bf16x4 hb = *reinterpret_cast<const bf16x4*>(&h);
#pragma unroll
            for (int e = 0; e < 4; ++e) hf[u >> 1][4 * (u & 1) + e] = hb[e];
            c16_slot_write(slot, t, kg, u, h);
        }
        Fs = stage_begin(S);
        c16_slot_flush(slot, p.dh1, p.F, 64 * c, row0, p.M, lane, nullptr, 0, 0);
        c16_stage(Fs, [&](int g, int i, const bf16x8& a) { const int j = (g & 1) * 8 + i; Z[j] = C16_MFMA(a, hf[g >> 1], Z[j]); },
                  [&](int g) { issue2(S + CH_NRING - 1, g); });
        ++S;
    }
    ch_lds_fence();
    if constexpr (SPLIT) {
        // ---- the partner's partial sum of g (chain16.h: the same hand-shake, wave by wave)
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
#pragma unroll
        for (int b8 = 0; b8 < 2; ++b8) {                 // (two round trips of eight tiles: this kernel has no 64 registers to spare)
            f32x4v O[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float* t4 = theirs + (2 * b8 + q) * 1024;
                c16_load16_sc<0>(O[4 * q], t4); c16_load16_sc<1024>(O[4 * q + 1], t4);
                c16_load16_sc<2048>(O[4 * q + 2], t4); c16_load16_sc<3072>(O[4 * q + 3], t4);
            }
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(O[0]), "+v"(O[1]), "+v"(O[2]), "+v"(O[3]), "+v"(O[4]), "+v"(O[5]), "+v"(O[6]), "+v"(O[7])
                         :
                         : "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) Z[8 * b8 + j][e] = side == 0 ? Z[8 * b8 + j][e] + O[j][e] : O[j][e] + Z[8 * b8 + j][e];   // P0 + P1
        }
        if (lane == 0) c16_store_flag_sc(p.xflag + oslot, 0u);
    }
    const bool wr = side == 0;                           // (wave-uniform) the first workgroup of a pair stores the row-wise results

    // ---- res-norm 1 backward on the wave's 16 rows (rowgemm.hip: rg_epilogue<1>, same arithmetic)
    bf16x8(&DO)[8] = DF;                                 // d_o's fragments take the registers of df's
    {
        const uint32_t rb = ttsmi_row_base(key, (uint32_t)rowc);
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 gm = *reinterpret_cast<const float4*>(gam + 16 * j + 4 * kg);
            const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = padded ? 0.f : Z[j][e];
                Z[j][e] = g;
                const float tt = g * gmv[e];
                s1[e] += tt;
                s2[e] += tt * ch_bf(XH[j >> 1], 4 * (j & 1) + e);
            }
        }
        const float invC = 1.0f / (float)CH_D;
        float m1 = (s1[0] + s1[1]) + (s1[2] + s1[3]), m2 = (s2[0] + s2[1]) + (s2[2] + s2[3]);
        m1 += __shfl_xor(m1, 16, 64); m1 += __shfl_xor(m1, 32, 64);
        m2 += __shfl_xor(m2, 16, 64); m2 += __shfl_xor(m2, 32, 64);
        m1 *= invC; m2 *= invC;
        // per 64 features: dz -> dh (residual gradient) and d_o = keep(dz) leave through the slot
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            uint2 dz_q[4], do_q[4];
            float dzf[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = 4 * cc + u, c0 = 16 * j + 4 * kg;
                const float4 gm = *reinterpret_cast<const float4*>(gam + c0);
                const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
                float dz[4], dx[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = ch_bf(XH[j >> 1], 4 * (j & 1) + e);
                    dz[e] = rstd * (Z[j][e] * gmv[e] - m1 - xh * m2);
                    dzf[u][e] = dz[e];
                }
                if (p.thr) {
                    const uint32_t h0 = ttsmi_pair_hash(rb, (uint32_t)c0), h1 = ttsmi_pair_hash(rb, (uint32_t)(c0 + 2));
                    dx[0] = dz[0] * (((h0 & 0xFFFFu) >= p.thr) ? p.inv_keep : 0.f);
                    dx[1] = dz[1] * (((h0 >> 16) >= p.thr) ? p.inv_keep : 0.f);
                    dx[2] = dz[2] * (((h1 & 0xFFFFu) >= p.thr) ? p.inv_keep : 0.f);
                    dx[3] = dz[3] * (((h1 >> 16) >= p.thr) ? p.inv_keep : 0.f);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dx[e] = dz[e];
                }
                dz_q[u] = ch_pack4(dz[0], dz[1], dz[2], dz[3]);
                do_q[u] = ch_pack4(dx[0], dx[1], dx[2], dx[3]);
                const bf16x4 hb = *reinterpret_cast<const bf16x4*>(&do_q[u]);
#pragma unroll
                for (int e = 0; e < 4; ++e) DO[j >> 1][4 * (j & 1) + e] = hb[e];      // (df's value at this slot is no longer needed)
            }
            if (!wr) continue;
            // d_o
#pragma unroll
            for (int u = 0; u < 4; ++u) c16_slot_write(slot, t, kg, u, do_q[u]);
            ch_lds_fence();
            c16_slot_flush(slot, p.d_o, CH_D, 64 * cc, row0, p.M, lane, nullptr, 0, 0);
            ch_lds_fence();
            // dh (bf16 when a lower block's res-norm 2 backward consumes it in the next GEMM's epilogue, fp32 otherwise)
            if (p.dres_bf16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) c16_slot_write(slot, t, kg, u, dz_q[u]);
                ch_lds_fence();
                c16_slot_flush(slot, reinterpret_cast<uint16_t*>(p.dres), CH_D, 64 * cc, row0, p.M, lane, nullptr, 0, 0);
                ch_lds_fence();
            } else {
#pragma unroll
                for (int hv = 0; hv < 2; ++hv) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        *reinterpret_cast<float4*>(slot + (t * 36 + 16 * u + 4 * kg) * 4) =
                            make_float4(dzf[2 * hv + u][0], dzf[2 * hv + u][1], dzf[2 * hv + u][2], dzf[2 * hv + u][3]);
                    ch_lds_fence();
                    c16_slot_flush_f32(slot, reinterpret_cast<float*>(p.dres), 64 * cc + 32 * hv, row0, p.M, lane);
                    ch_lds_fence();
                }
            }
        }
        // ---- parameter-gradient partials.  Column sums of g x^ and g over the wave's 16 rows: the 16 lanes of a DPP row hold
        // the 16 rows of one feature group, so four row-local DPP steps (common.h: the first four of wave_sum_dpp) leave the
        // sum in every lane - no LDS, fixed order.  Lane t == 0 of each group writes them to the wave's slot as
        // [dgamma 256 | dbeta 256] floats; after the next stage's barrier every wave adds 64 of the 512 columns over the eight
        // slots in wave order and stores them into the workgroup's partial row (ttsmi_layernorm_param_reduce_batched_nw layout).
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float sg[4], sb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float vg = Z[j][e] * ch_bf(XH[j >> 1], 4 * (j & 1) + e), vb = Z[j][e];
                vg += ttsmi_dpp<TTSMI_DPP_QUAD_1032, 0xF>(vg, 0.f);           vb += ttsmi_dpp<TTSMI_DPP_QUAD_1032, 0xF>(vb, 0.f);
                vg += ttsmi_dpp<TTSMI_DPP_QUAD_2301, 0xF>(vg, 0.f);           vb += ttsmi_dpp<TTSMI_DPP_QUAD_2301, 0xF>(vb, 0.f);
                vg += ttsmi_dpp<TTSMI_DPP_ROW_HALF_MIRROR, 0xF>(vg, 0.f);     vb += ttsmi_dpp<TTSMI_DPP_ROW_HALF_MIRROR, 0xF>(vb, 0.f);
                vg += ttsmi_dpp<TTSMI_DPP_ROW_MIRROR, 0xF>(vg, 0.f);          vb += ttsmi_dpp<TTSMI_DPP_ROW_MIRROR, 0xF>(vb, 0.f);
                sg[e] = vg; sb[e] = vb;
            }
            if (t == 0) {
                float* sf = reinterpret_cast<float*>(slot);
                *reinterpret_cast<float4*>(sf + 16 * j + 4 * kg) = make_float4(sg[0], sg[1], sg[2], sg[3]);
                *reinterpret_cast<float4*>(sf + CH_D + 16 * j + 4 * kg) = make_float4(sb[0], sb[1], sb[2], sb[3]);
            }
        }
    }

    // ---- dctx = d_o . Wo[d:2d]^T: 4 stages of (4 output tiles x 8 k-blocks)
    const int q0 = side * nq_w;                          // (SPLIT: this workgroup's half of the 256 columns)
    for (int s_ = 0; s_ < nq_w; ++s_) {
        const int s = q0 + s_;
        const unsigned char* Fs = stage_begin(S);
        if (s_ == 0) {
            // (the barrier above published every wave's column sums)
#pragma unroll
            for (int cpart = 0; cpart < 8 / NW; ++cpart) {
                const int col = 64 * (wave * (8 / NW) + cpart) + lane;     // 0..511: dgamma columns, then dbeta columns
                float a = 0.f;
#pragma unroll
                for (int wv = 0; wv < NW; ++wv) a += reinterpret_cast<const float*>(scr + wv * C16_SLOT_BYTES)[col];
                if (wr && tile < p.nparts) p.part[((long)(col >> 8) * p.nparts + tile) * CH_D + (col & 255)] = a;
            }
            ch_barrier();                                              // the slots are free again
        } else {
            c16_slot_flush(slot, p.dctx, CH_D, 64 * (s - 1), row0, p.M, lane, nullptr, 0, 0);
        }
        f32x4v acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[u][e] = 0.f;
        c16_stage(Fs, [&](int g, int i, const bf16x8& a) { acc[i & 3] = C16_MFMA(a, DO[2 * g + (i >> 2)], acc[i & 3]); },
                  [&](int g) { issue2(S + CH_NRING - 1, g); });
        ++S;
        ch_lds_fence();
#pragma unroll
        for (int u = 0; u < 4; ++u) c16_slot_write(slot, t, kg, u, ch_pack4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]));
    }
    ch_lds_fence();
    c16_slot_flush(slot, p.dctx, CH_D, 64 * (q0 + nq_w - 1), row0, p.M, lane, nullptr, 0, 0);
}
