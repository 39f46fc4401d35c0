// wav -> reflect-padded frames -> periodic-Hann window -> 1024-pt real FFT -> |.| -> sparse Slaney
// mel filterbank -> clip + log, as ONE batched kernel (data/audio.py:72-92,209-242).
//
// One wave64 owns one frame.  The n_fft-point real FFT is an (n_fft/2)-point complex FFT of
// z[n] = x[2n] + i x[2n+1].  n_fft = 1024 (MelGAN / LJSpeech config): one 512-point transform done as
// three radix-8 Stockham passes (512 = 8^3).  n_fft = 2048 (WaveRNN config, data_config_wavernn.yaml): a
// radix-2 decimation-in-frequency stage in registers splits the 1024 points into two 512-point
// transforms (even / odd output bins) that run through the same three passes.  In each 512-point transform each lane keeps 8 complex points in registers
// per pass, the two inter-pass exchanges go through a per-wave 4 KB LDS buffer, twiddles come from
// an LDS table of exp(-2 pi i k / 1024) built once per workgroup with sincospi.  The magnitude
// spectrum (513 bins) stays in LDS; the mel filterbank is applied in its sparse form (each filter
// is one contiguous run of bins, 727 non-zeros for the LJSpeech setting).  Round 6 (1 447 -> 1 610 GB/s on one box,
// profiles/r06_mel_scan_bpermute_ab.txt): the n_fft = 1024 transform's last pass stays in registers and the real-FFT
// post-processing fetches each mirror bin from its partner lane (ds_bpermute) instead of a third exchange through LDS; a
// filter's <= 8 work items sit in adjacent lanes and are summed by a segmented DPP scan, the lane with a filter's last item
// writes its output (the second sweep over partial sums in LDS is gone); the factor 1/2 of |X| lives in the weights.  HBM traffic: every sample is fetched ~once
// (the 4x frame overlap is served by L2 because consecutive frames run in the same workgroup),
// 4*n_mels bytes written per frame.
#include <stdlib.h>

#include "common.h"
#include "fft512.h"

#define FR_PER_WG 4     // waves per workgroup = frames in flight
#define MELS_MAX 128
#define MEL_IT 12         // weights per mel work item
struct MelP {
    const float* wav; const int64_t* clip_off; const int64_t* frame_off;
    int n_clips; long total_frames; int hop;
    const float* window;
    int n_mels; const int32_t* mel_lo; const int32_t* mel_cnt; const int32_t* mel_ptr;
    const float* mel_w;
    int normalizer; float clip_min;
    float* out;
    int groups_per_wg;
    int ablate;      // measurement only (-DTTSMI_ABLATION_BUILD + TTSMI_MEL_ABLATE): 1 = no sample loads, 2 = no mel stage, 4 = no passes 2/3, 8 = no post-processing
};

__device__ __forceinline__ int clip_of_frame(const int64_t* frame_off, int n_clips, long f) {
    int lo = 0, hi = n_clips;        // find c with frame_off[c] <= f < frame_off[c+1]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (frame_off[mid] <= f) lo = mid; else hi = mid;
    }
    return lo;
}

template <int NFFT>
__global__ __launch_bounds__(256, NFFT == 1024 ? 4 : 1) void stft_logmel_kernel(MelP p) {
    constexpr int NC = NFFT / 2;                // complex points
    constexpr int NSUB = NC / SUBN;             // 512-point sub-transforms per frame (1 or 2)
    constexpr int PPL = NC / 64;                // complex points per lane
    constexpr int MEL_ITEMS = NFFT == 1024 ? 128 : 256;   // LDS item-table capacity (LJSpeech bank: 105 items)
    constexpr int MELW_MAX = NFFT == 1024 ? 1536 : 2304;  // LDS copy of the sparse weights (<= 2 per bin)
    // twiddles.  NSUB == 1 (n_fft 1024): the three passes only index the EVEN entries of exp(-2 pi i k / NFFT) = the table
    // of the 512-point transform itself (tw, NC entries), the real-FFT post-processing the first NC / 2 of the full table
    // (twp): 6 KB instead of 8 - with the set-up arrays below living in the exchange buffers that brings a workgroup to
    // 40 KB of LDS, FOUR per CU.  NSUB == 2: the full table (its radix-2 stage and post-processing index all of it).
    constexpr int TWN = NSUB == 1 ? NC : NFFT;          // entries / period of tw
    __shared__ float2 tw[TWN];
    __shared__ float2 twp[NSUB == 1 ? NC / 2 : 1];
    __shared__ float2 buf[FR_PER_WG][NSUB][ZBUF];     // padded: physical index = i + (i >> 3)
    __shared__ __attribute__((aligned(16))) float mag[FR_PER_WG][NC + 8 + MEL_IT];      // bins NC+1.. stay 0 (mel items may read past NC)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int k = tid; k < TWN; k += 256) {
        float s, c;
        sincospif(-2.0f * (float)k / (float)TWN, &s, &c);
        tw[k] = make_float2(c, s);
    }
    if constexpr (NSUB == 1) {
        for (int k = tid; k < NC / 2; k += 256) {
            float s, c;
            sincospif(-2.0f * (float)k / (float)NFFT, &s, &c);
            twp[k] = make_float2(c, s);
        }
    }
    // sparse filterbank -> LDS once per workgroup (727 weights + 3 x 80 ints for the LJSpeech setting);
    // larger banks than the LDS copy holds are read from global memory instead
    // (set-up only: the copy lives in the exchange buffers, which the first transform overwrites)
    static_assert(sizeof(float) * MELW_MAX + 3 * sizeof(int) * MELS_MAX <= sizeof(buf), "the filterbank copy lives in buf");
    float* melwS = reinterpret_cast<float*>(&buf[0][0][0]);
    int* melloS = reinterpret_cast<int*>(melwS + MELW_MAX);
    int* melcntS = melloS + MELS_MAX;
    int* melptrS = melcntS + MELS_MAX;
    const int nnz = p.mel_ptr[p.n_mels - 1] + p.mel_cnt[p.n_mels - 1];
    const bool mel_in_lds = (nnz <= MELW_MAX) && (p.n_mels <= MELS_MAX);
    if (mel_in_lds) {
        for (int i = tid; i < nnz; i += 256) melwS[i] = p.mel_w[i];
        for (int i = tid; i < p.n_mels; i += 256) {
            melloS[i] = p.mel_lo[i]; melcntS[i] = p.mel_cnt[i]; melptrS[i] = p.mel_ptr[i];
        }
    }
    __syncthreads();
    // Lane balance of the sparse mel product: the Slaney filters are 3 bins wide at the bottom and ~37 at
    // the top, so "one lane per filter" leaves most of the wave idle behind the widest filter.  Filters
    // are cut into work items of at most MEL_IT weights (about 105 items for the LJSpeech bank); a lane
    // sums one item per round, a second short pass adds the (<= 4) partial sums of each filter.
    // (the item tables that only the set-up reads - where an item's magnitudes start, its place in its filter, the filters'
    // first slots - live behind the filterbank copy in the exchange buffers; every lane keeps its own entries in registers)
    static_assert(sizeof(float) * MELW_MAX + sizeof(int) * (3 * MELS_MAX + 2 * MEL_ITEMS + MELS_MAX + 2) <= sizeof(buf), "the set-up tables live in buf");
    int* itLo = melptrS + MELS_MAX;
    int* itMeta = itLo + MEL_ITEMS;
    int* itFirst = itMeta + MEL_ITEMS;
    int& itMaxN = itFirst[MELS_MAX + 1];
    __shared__ __attribute__((aligned(16))) float itWt[MEL_ITEMS][MEL_IT];   // zero-padded: fixed trip count
    // Round 6: a filter's items sit in ADJACENT lanes of one 16-lane DPP row (a filter that would straddle a row starts at the
    // next one; the skipped slots multiply zeros), so a filter's sum is a segmented scan over row_shr 1 / 2 / 4 in registers and
    // the lane that holds a filter's LAST item finishes it (clip, log, store).  The partial sums used to go through LDS to a
    // second sweep - a loop of data-dependent length over them, two more table reads per filter and a wave fence: ~100 of a
    // frame's ~750 instructions.  itMeta: bits 0-7 = the item's index within its filter, bits 8.. = filter + 1 on its last item.
    for (int i = tid; i < MEL_ITEMS * MEL_IT; i += 256) (&itWt[0][0])[i] = 0.f;
    for (int i = tid; i < MEL_ITEMS; i += 256) { itLo[i] = 0; itMeta[i] = 0; }
    if (mel_in_lds) {
        if (tid == 0) {
            // an item starts at a multiple of 4 bins (the filter's first bin rounded down, zero weights in front): its
            // magnitudes are read as three aligned float4 instead of twelve scalars
            int acc = 0, most = 0;
            for (int m = 0; m < p.n_mels; ++m) {
                const int n = ((melloS[m] & 3) + melcntS[m] + MEL_IT - 1) / MEL_IT;
                if ((acc & 15) + n > 16) acc = (acc + 15) & ~15;
                itFirst[m] = acc;
                acc += n;
                most = n > most ? n : most;
            }
            itFirst[p.n_mels] = acc;
            itMaxN = most;
        }
        __syncthreads();
        if (itFirst[p.n_mels] <= MEL_ITEMS && itMaxN <= 8) {
            for (int m = tid; m < p.n_mels; m += 256) {
                const int first = itFirst[m];
                const int lead = melloS[m] & 3, lo4 = melloS[m] - lead;
                const int n = (lead + melcntS[m] + MEL_IT - 1) / MEL_IT;
                for (int j = 0; j < n; ++j) {
                    itLo[first + j] = lo4 + j * MEL_IT;
                    itMeta[first + j] = j | (j == n - 1 ? (m + 1) << 8 : 0);
                    for (int i = 0; i < MEL_IT; ++i) {
                        const int wi = j * MEL_IT + i - lead;                   // index into the filter's weights
                        itWt[first + j][i] = (wi >= 0 && wi < melcntS[m]) ? 0.5f * melwS[melptrS[m] + wi] : 0.f;   // (x 1/2: mag holds 2 |X|)
                    }
                }
            }
        }
        __syncthreads();
    }
    const int n_items = mel_in_lds ? itFirst[p.n_mels] : 0;          // item slots, row padding included
    const bool mel_items = mel_in_lds && n_items <= MEL_ITEMS && itMaxN <= 8;
    // the lane's item of every round never changes: where its magnitudes start, its place in its filter
    constexpr int MEL_ROUNDS = MEL_ITEMS / 64;
    int it_lo[MEL_ROUNDS], it_meta[MEL_ROUNDS];
#pragma unroll
    for (int rd = 0; rd < MEL_ROUNDS; ++rd) {
        it_lo[rd] = mel_items ? itLo[(tid & 63) + 64 * rd] : 0;
        it_meta[rd] = mel_items ? itMeta[(tid & 63) + 64 * rd] : 0;
    }
    // last bin any filter reads (filters are stored in ascending order of their first bin; without the LDS copy: all)
    int bin_hi = NC;
    if (mel_in_lds) {
        bin_hi = 0;
        for (int m = 0; m < p.n_mels; ++m) bin_hi = max(bin_hi, melloS[m] + melcntS[m] - 1);
        bin_hi = __builtin_amdgcn_readfirstlane(bin_hi);
    }
    __syncthreads();                                  // the last readers of the filterbank copy: buf is the transforms' from here

    float2* zb = buf[wave][0];
    float* mg = mag[wave];
    for (int i = lane; i < NC + 8 + MEL_IT; i += 64) mg[i] = 0.f;       // bins no round writes (above the bank, the padding) read as 0
    // the lane's 2 * PPL window taps never change: keep them in registers
    float2 win[PPL];
#pragma unroll
    for (int r = 0; r < PPL; ++r) {
        int n = 2 * (lane + 64 * r);
        win[r] = make_float2(p.window[n], p.window[n + 1]);
    }
    // clip of the previous frame: a wave's consecutive frames rarely change clip
    int cur = -1;
    long cur_f0 = 0, cur_f1 = 0, cur_base = 0;
    int cur_L = 1;
    // Raw samples of frame f -> x[PPL] (pairs z[i + 64 r]), requested one iteration ahead of their use so that the HBM / L2
    // latency of frame g+1 hides under the FFT of frame g.  The prefetch is ALWAYS the reflection-free pattern (from a
    // clamped, in-bounds address) and is only kept for interior frames; the ~4 edge frames of a clip are loaded with
    // reflection when they are consumed.  (Round 1/2 chose between the two patterns in the prefetch: the merge of the two
    // paths copied the loaded registers, i.e. `s_waitcnt vmcnt(0)` right behind the prefetch - nothing overlapped.)
    const long total_samples = p.clip_off[p.n_clips];
    const bool can_prefetch = total_samples >= NFFT;          // (always, except on toy inputs)
    struct Pending { bool act, interior; long base; int L, start; } nx = {false, true, 0, 1, 0};
    auto prefetch_frame = [&](long f, float2 (&x)[PPL]) {
        nx.act = (f < p.total_frames) && !(TTSMI_ABLATE_BITS(p.ablate) & 1);
        nx.base = 0; nx.L = 1;
        int t = 0;
        if (nx.act) {
            if (cur < 0 || f < cur_f0 || f >= cur_f1) {   // wave-uniform: one dependent search per clip change
                cur = clip_of_frame(p.frame_off, p.n_clips, f);
                cur_f0 = p.frame_off[cur];
                cur_f1 = p.frame_off[cur + 1];
                cur_base = p.clip_off[cur];
                cur_L = (int)(p.clip_off[cur + 1] - cur_base);
            }
            nx.base = cur_base;
            nx.L = cur_L;
            t = (int)(f - cur_f0);
        }
        nx.start = t * p.hop - NFFT / 2;
        nx.interior = nx.act && nx.start >= 0 && nx.start + NFFT <= nx.L && can_prefetch;
        // UNCONDITIONAL loads (round 6): under `if (total_samples >= NFFT)` and `if (g + 1 < groups)` the prefetched
        // registers were a phi of "loaded" and "kept" - 16 v_mov_b64 per frame around the loads (ISA reading).  A toy input
        // shorter than one frame prefetches from the window table instead (n_fft floats, always there) and loads every
        // frame in fix_frame.
        long s0 = nx.base + nx.start;
        s0 = s0 < 0 ? 0 : (s0 > total_samples - NFFT ? total_samples - NFFT : s0);
        const float* src = (can_prefetch ? p.wav + s0 : p.window) + 2 * lane;
#pragma unroll
        for (int r = 0; r < PPL; ++r) x[r] = make_float2(src[128 * r], src[128 * r + 1]);
    };
    // at consumption: an edge frame (or an inactive slot) replaces the speculative prefetch
    auto fix_frame = [&](float2 (&x)[PPL]) {
        if (nx.interior) return;                          // wave-uniform
#pragma unroll
        for (int r = 0; r < PPL; ++r) {
            int n = 2 * (lane + 64 * r);
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int gi = nx.start + n + e;
                if (gi < 0) gi = -gi;                     // np.pad(mode='reflect')
                if (gi >= nx.L) gi = 2 * (nx.L - 1) - gi;
                gi = gi < 0 ? 0 : gi;
                v[e] = nx.act ? p.wav[nx.base + gi] : 0.f;
            }
            x[r] = make_float2(v[0], v[1]);
        }
    };
    const long f_first = (long)blockIdx.x * p.groups_per_wg * FR_PER_WG + wave;
    float2 xs[PPL];
#pragma unroll
    for (int r = 0; r < PPL; ++r) xs[r] = make_float2(0.f, 0.f);
    prefetch_frame(f_first, xs);
    for (int g = 0; g < p.groups_per_wg; ++g) {
        const long f = f_first + (long)g * FR_PER_WG;
        const bool active = f < p.total_frames;
        fix_frame(xs);
        float2 w[PPL];
#pragma unroll
        for (int r = 0; r < PPL; ++r) w[r] = make_float2(xs[r].x * win[r].x, xs[r].y * win[r].y);
        prefetch_frame(f + FR_PER_WG, xs);                 // (past the last group: an inactive frame, a clamped in-bounds address)
        const bool full = !(TTSMI_ABLATE_BITS(p.ablate) & 4);
        if constexpr (NSUB == 1) {
            fft512<NC, false>(w, zb, tw, lane, full);      // Z[lane + 64 r] in w[rev3(r)]
        } else {
            // radix-2 DIF stage in registers: z[n] and z[n + 512] sit in the same lane (r and r + 8);
            //   even bins Z[2k]   = FFT512(z[n] + z[n+512])
            //   odd  bins Z[2k+1] = FFT512((z[n] - z[n+512]) * exp(-2 pi i n / 1024))
            float2 ua[8], ub[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                ua[r] = cadd(w[r], w[r + 8]);
                ub[r] = cmul(csub(w[r], w[r + 8]), tw[2 * (lane + 64 * r)]);   // exp(-2 pi i n/1024) = tw2048[2n]
            }
            fft512<NFFT>(ua, buf[wave][0], tw, lane, full);
            fft512<NFFT>(ub, buf[wave][NSUB - 1], tw, lane, full);
        }
        auto Z = [&](int k) -> float2 {                    // k in [0, NC)
            if constexpr (NSUB == 1) return zb[ZP(k)];
            else return buf[wave][k & 1][ZP(k >> 1)];
        };
        // ---- real-FFT post-processing: |X[k]|, k = 0..NC ---------------------------------------
        // X[k] = (e - i w o)/2 with e = Z[k] + conj(Z[NC-k]), o = Z[k] - conj(Z[NC-k]), w = exp(-2 pi i k/NFFT);
        // the mirror bin shares everything: X[NC-k] = (conj(e) - i conj(w o))/2.  One pass over k = 0..NC/2
        // yields both magnitudes (half the LDS reads and complex multiplies of a pass over all bins).
        // (v_sqrt_f32 is accurate to 1 ulp; the library sqrtf adds a denormal rescale and a correctly-rounded fix-up -
        // ~15 instructions each, 10 square roots per lane and frame were a sixth of the kernel's VALU work.  Bins above
        // the highest filter's last bin - 371 of 512 for the LJSpeech bank - feed nothing: their mirror is skipped.)
        // NSUB == 1 (round 6): the transform's last pass leaves Z[lane + 64 r] in the lane's registers, and the mirror of
        // bin lane + 64 r is register 7 - r of lane 64 - lane: one cross-lane read (ds_bpermute, no LDS bytes) per component
        // instead of eight 8-byte stores, a fence and eight loads.  Lane 0 is its own partner with the registers shifted by
        // one (the mirror of bin 64 r is bin 64 (8 - r)): two selects per round.
        const int rev3[8] = {0, 4, 2, 6, 1, 5, 3, 7};
        const int partner = ((64 - lane) & 63) << 2;
#pragma unroll
        for (int it = 0; it < NC / 128; ++it) {                 // k = 0 .. NC/2 - 1 in full waves; k = NC/2 below
            if (TTSMI_ABLATE_BITS(p.ablate) & 8) break;
            const int k = lane + 64 * it;
            float2 zk, zc;
            if constexpr (NSUB == 1) {
                zk = w[rev3[it]];
                const float2 src = w[rev3[7 - it]], own = w[rev3[(8 - it) & 7]];
                zc.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner, __builtin_bit_cast(int, src.x)));
                zc.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner, __builtin_bit_cast(int, src.y)));
                zc.x = lane == 0 ? own.x : zc.x;
                zc.y = lane == 0 ? own.y : zc.y;
            } else {
                zk = Z(k);
                zc = Z((NC - k) & (NC - 1));
            }
            zc.y = -zc.y;
            float2 e = cadd(zk, zc), o = csub(zk, zc);
            float2 wo = cmul(NSUB == 1 ? twp[k] : tw[k], o);
            float xr = e.x + wo.y, xi = e.y - wo.x;            // 2 X[k]
            mg[k] = __builtin_amdgcn_sqrtf(xr * xr + xi * xi);          // 2 |X[k]|: the exact factor 1/2 lives in the filter weights
            if (NC - 64 * it - 63 > bin_hi + MEL_IT) continue;  // wave-uniform: no mirror bin of this round is read (they stay 0)
            float yr = e.x - wo.y, yi = e.y + wo.x;            // 2 conj-mirrored X[NC-k]
            mg[NC - k] = __builtin_amdgcn_sqrtf(yr * yr + yi * yi);
        }
        if (lane == 0 && !(TTSMI_ABLATE_BITS(p.ablate) & 8)) {                    // k = NC/2 is its own mirror: X = conj(Z)
            const float2 zh = NSUB == 1 ? w[rev3[(NC / 128) & 7]] : Z(NC / 2);      // (NSUB == 1: lane 0's register 4 = Z[256])
            mg[NC / 2] = 2.f * __builtin_amdgcn_sqrtf(zh.x * zh.x + zh.y * zh.y);
        }
        WAVE_SYNC();
        // ---- sparse mel + normalisation ---------------------------------------------------------
        auto finish = [&](float sum) -> float {
            if (p.normalizer == 0) return __logf(fmaxf(sum, p.clip_min));   // v_log_f32 (1 ulp in log2) x ln 2; the argument is >= clip_min > 0
            const float db = 20.f * log10f(fmaxf(1e-5f, sum));
            const float nz = fminf(fmaxf((db + 100.f) / 100.f, 0.f), 1.f);
            return nz * 8.f - 4.f;
        };
        if (mel_items && !(TTSMI_ABLATE_BITS(p.ablate) & 2)) {
            // The products are formed as PAIRS along the weight index (v_pk_fma_f32 on the two halves of each 16-byte read):
            // written with four scalar accumulators, hipcc's SLP pass paired the SAME accumulator of two loop iterations
            // instead and spent 36 v_mov_b32 per 12 packed FMAs shuffling the operands together (ISA reading, round 4).
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int rd = 0; rd < MEL_ROUNDS; ++rd) {
                if (64 * rd >= n_items) break;                                  // wave-uniform
                const f32x4_t* w4 = reinterpret_cast<const f32x4_t*>(itWt[lane + 64 * rd]);
                const f32x4_t* x4 = reinterpret_cast<const f32x4_t*>(mg + it_lo[rd]);   // 16-byte aligned; may run past the filter: zero weights
                f32x2_t a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < MEL_IT / 4; ++i) {
                    const f32x4_t w = w4[i], x = x4[i];
                    a0 += w.xy * x.xy;
                    a1 += w.zw * x.zw;
                }
                a0 += a1;
                float a = a0.x + a0.y;
                // segmented inclusive scan over the filter's (<= 8) adjacent items: row_shr never leaves the 16-lane row, and
                // an item adds its d-th left neighbour only when that one belongs to the same filter
                const int pos = it_meta[rd] & 0xFF, mel = (it_meta[rd] >> 8) - 1;
                float v = ttsmi_dpp<0x111, 0xF>(a, 0.f);
                a += pos >= 1 ? v : 0.f;
                v = ttsmi_dpp<0x112, 0xF>(a, 0.f);
                a += pos >= 2 ? v : 0.f;
                v = ttsmi_dpp<0x114, 0xF>(a, 0.f);
                a += pos >= 4 ? v : 0.f;
                if (mel >= 0 && active) p.out[f * p.n_mels + mel] = finish(a);
                __builtin_amdgcn_sched_barrier(0);                              // one round's twelve reads in flight at a time
            }
        } else if (!(TTSMI_ABLATE_BITS(p.ablate) & 2)) {
            for (int m = lane; m < p.n_mels; m += 64) {   // a bank beyond the item table: straight from global memory
                const int lo = p.mel_lo[m], cnt = p.mel_cnt[m];
                const float* w = p.mel_w + p.mel_ptr[m];
                float sum = 0.f;
                for (int i = 0; i < cnt; ++i) sum += w[i] * mg[lo + i];
                if (active) p.out[f * p.n_mels + m] = finish(0.5f * sum);
            }
        }
        WAVE_SYNC();
    }
}

extern "C" {

int ttsmi_stft_logmel(const float* wav, const int64_t* clip_off, const int64_t* frame_off,
                      int n_clips, int64_t total_frames, int n_fft, int hop, const float* window,
                      int n_mels, const int32_t* mel_lo, const int32_t* mel_cnt,
                      const int32_t* mel_ptr, const float* mel_w, int normalizer, float clip_min,
                      float* out, ttsmi_stream_t stream) {
    TTSMI_CHECK_ARG(wav && clip_off && frame_off && window && mel_lo && mel_cnt && mel_ptr &&
                        mel_w && out, "stft_logmel: null pointer");
    TTSMI_CHECK_ARG(n_clips > 0 && total_frames > 0 && hop > 0 && n_mels > 0,
                    "stft_logmel: bad shape");
    if (n_fft != 1024 && n_fft != 2048) {
        ttsmi_set_error("stft_logmel: n_fft=%d not built (1024 and 2048 are)", n_fft);
        return TTSMI_ERR_UNSUPPORTED;
    }
    TTSMI_CHECK_ARG(normalizer == 0 || normalizer == 1, "stft_logmel: unknown normalizer %d", normalizer);
    MelP p;
    p.wav = wav; p.clip_off = clip_off; p.frame_off = frame_off; p.n_clips = n_clips;
    p.total_frames = total_frames; p.hop = hop; p.window = window; p.n_mels = n_mels;
    p.mel_lo = mel_lo; p.mel_cnt = mel_cnt; p.mel_ptr = mel_ptr; p.mel_w = mel_w;
    p.normalizer = normalizer; p.clip_min = clip_min; p.out = out;
    TTSMI_ABLATE_KNOB(ablate, "TTSMI_MEL_ABLATE");      // wrong results by construction: only in a measurement build
    p.ablate = ablate;
    long groups = (total_frames + FR_PER_WG - 1) / FR_PER_WG;
    // A workgroup's set-up (1024 sincospi, the filterbank copy, a SERIAL 80-step prefix over the filters, the weight
    // item table) costs ~10 us: with 64 frames per workgroup (round 1/2: gpw <= 16) that was ~13 % of a 10 000-clip
    // launch.  Few, long-lived workgroups instead: ~64 per CU (4 resident, ~16 rounds: measured best of 4 096 .. 24 576 at four
    // resident - 1 327 / 1 386 / 1 416 / 1 410 GB/s at 4 096 / 8 192 / 16 384 / 24 576), each a contiguous range of
    // frames, so the set-up is amortised over hundreds of frames.  TTSMI_MEL_WGS overrides the target (A/B knob).
    TTSMI_KNOB(target_env, "TTSMI_MEL_WGS", 16384);
    const long target = target_env > 0 ? target_env : 16384;
    long gpw_l = (groups + target - 1) / target;
    if (gpw_l < 1) gpw_l = 1;
    if (gpw_l > (1 << 20)) gpw_l = 1 << 20;
    int gpw = (int)gpw_l;
    p.groups_per_wg = gpw;
    long blocks = (groups + gpw - 1) / gpw;
    if (n_fft == 1024)
        hipLaunchKernelGGL(stft_logmel_kernel<1024>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(stft_logmel_kernel<2048>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    TTSMI_CHECK_LAUNCH("stft_logmel");
    return TTSMI_OK;
}

}  // extern "C"
