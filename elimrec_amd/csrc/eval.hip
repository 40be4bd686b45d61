// Full-catalogue counterfactual scoring (TE / TIE), train-item masking, top-K and the ranking
// metrics, all on device (models/EliMRec.py:96-113,155-212; cpp/uni_evaluator.py:131-185;
// evaluate.h:23-42; metric.h:17-106). Removes the [B_t x I] device->host copy of the reference.
#include "common.h"
#include <cstdlib>

namespace elimrec {

constexpr int kMaxS = 4;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// reductions over the 16 lanes of a DPP row (the 16 items of a tile) on the VALU: quad swaps, half-row mirror, row mirror;
// every lane of the row ends with the result (HIP's __shfl_xor is an LDS-pipe ds_bpermute per step)
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    return fmaxf(v, dpp_f<0x140>(v));
}
// row16_max for values that are non-negative or -inf (scores: sigmoids, -inf beyond the chunk): their bit patterns order as
// signed integers the way the floats do, and an integer max needs no NaN canonicalisation around every step and takes the DPP
// operand itself (4 instructions per row instead of 15) -- the same bits as row16_max
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ float row16_max_nonneg(float f) {
    int v = __builtin_bit_cast(int, f);
    v = max(v, dpp_i<0xB1>(v));
    v = max(v, dpp_i<0x4E>(v));
    v = max(v, dpp_i<0x141>(v));
    v = max(v, dpp_i<0x140>(v));
    return __builtin_bit_cast(float, v);
}
__device__ __forceinline__ float row16_sum(float v) {      // fixed order: ((a+b)+(c+d)) quads, then the mirrored halves
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    return v + dpp_f<0x140>(v);
}

// Evaluation math (elimrec_score_set_math). EXACT (0) = the expressions above with IEEE division and libm expf -- about 196
// VALU instructions per (user, item) pair, which is what bounds the scorer then (58 % VALU-issue busy against 31 % MFMA busy;
// FAST: 47 % against 40 %).
// FAST (1, the default) = sigmoids through v_exp_f32 with a two-float argument product and v_rcp_f32 + one Newton step, and
// reciprocal norms refined the same way: every factor within ~2 ulp of the EXACT form, scores within 1.2e-7 absolute
// (tools/eval_math_accuracy.py, tests/test_hip_parity.py), a validation pass 0.027 s against 0.033. libm's expf is itself
// a <= 1 ulp approximation and not the one the reference's torch build uses, so neither mode is "the" reference bit
// pattern; both sit inside the 1e-5 the predict parity tests allow by two orders of magnitude.
// 1 / d to <= 1 ulp: v_rcp_f32 + one Newton step
__device__ __forceinline__ float rcp_nr(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.f), r, r);
}
// exp(-x) through v_exp_f32 with the product x * log2(e) carried in two floats: the rounding of the product would
// otherwise cost |x| * 2^-24 relative (the 2e-6 of the first FAST form at |x| ~ 16); with the residual applied as
// 2^err = 1 + err * ln 2 the result is within ~2 ulp of libm's for |x| < 80
__device__ __forceinline__ float exp_neg_(float x) {
    const float c_hi = 1.44269502162933349609375f, c_lo = 1.92596299112661746e-8f;   // log2(e) = c_hi + c_lo
    const float xc = fminf(fmaxf(x, -88.f), 100.f);        // exp stays finite: sigmoid saturates to 2.7e-39 / 1 beyond
    const float t = -xc * c_hi;
    const float err = fmaf(-xc, c_lo, fmaf(-xc, c_hi, -t));
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e * err, 0.693147180559945309f, e);
}
template <bool FAST> __device__ __forceinline__ float sig_(float x) {
    if (FAST) return rcp_nr(1.f + exp_neg_(x));
    return sigmoidf_(x);
}
// FAST forms for BOUNDED arguments -- the heads' z_h are cosines of normalised rows (|z| <= 1), ui / the row mean / the rubi
// fusion are sigmoids or products of sigmoids (in (0, 1)), a difference of two rubi fusions is in (-1, 1): the rounding of
// x * log2(e) costs |x| 2^-24 relative there, below the final rounding, so the two-float product and the clamp of exp_neg_
// are not needed (one multiply + v_exp_f32 instead of six operations + v_exp_f32) ...
__device__ __forceinline__ float exp_neg_small_(float x) { return __builtin_amdgcn_exp2f(x * -1.44269502162933349609375f); }
template <bool FAST> __device__ __forceinline__ float sig_small_(float x) {
    if (FAST) return rcp_nr(1.f + exp_neg_small_(x));
    return sigmoidf_(x);
}
// Where only the ABSOLUTE accuracy of a sigmoid matters -- ui = sigmoid(u . i) (used as a factor, a summand or averaged: never
// under a logarithm) and the last sigmoid of a score -- the product's rounding is harmless for any argument: the error of
// sigmoid is s (1 - s) |x| 2^-24 <= 1.4e-8. One clamp (v_exp_f32 must not overflow: 1 + inf -> NaN in the Newton step).
template <bool FAST> __device__ __forceinline__ float sig_abs_(float x) {
    if (FAST) return rcp_nr(1.f + __builtin_amdgcn_exp2f(fmaxf(x, -88.f) * -1.44269502162933349609375f));
    return sigmoidf_(x);
}
// ... and a sigmoid that is only ever AVERAGED over a catalogue (pass 1: the mean of sigmoid(u.i) behind the NDE term,
// models/EliMRec.py:107) needs neither the Newton step -- v_rcp_f32 is within 1 ulp, 3e-8 absolute on a value near 0.5, and the
// mean of 76 k such terms moves a TIE score by less than 1e-8 -- nor the clamp (1 + inf -> 0 without the Newton step)
template <bool FAST> __device__ __forceinline__ float sig_mean_(float x) {
    if (FAST) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.44269502162933349609375f));
    return sigmoidf_(x);
}
// ... and a PRODUCT of sigmoids goes through one reciprocal: prod_h 1 / (1 + e^-z_h) = 1 / prod_h (1 + e^-z_h), the
// denominator <= (1 + e)^4 (two v_rcp_f32 + Newton steps fewer per (user, item) pair with three heads)
template <bool MASKED> __device__ __forceinline__ float sig_den_(const float *z, int S, uint32_t mask) {
    float den = 1.f;
#pragma unroll
    for (int h = 0; h < 4; ++h)
        if (h < S && (!MASKED || (mask & (1u << h)))) den *= 1.f + exp_neg_small_(z[h]);
    return den;
}
// (the logarithms of the 'hm' / 'sum' fusions are libm's in both modes: v_log_f32 costs 7e-7 relative there)
template <bool FAST> __device__ __forceinline__ float log_(float x) { return logf(x); }
template <bool FAST> __device__ __forceinline__ float log1p_(float x) { return log1pf(x); }

// x: ui = sigmoid(u . i) (or the row mean of it), in (0, 1); z: the heads' cosines
template <bool FAST>
__device__ __forceinline__ float fuse_t(int mode, float x, const float *z, int S, uint32_t mask) {
    if (mode == 0) {
        if (FAST) return x * rcp_nr(sig_den_<true>(z, S, mask));
        float r = x;
#pragma unroll
        for (int h = 0; h < 4; ++h) if (h < S && (mask & (1u << h))) r *= sig_<FAST>(z[h]);
        return r;
    } else if (mode == 1) {
        float t;
        if (FAST) t = rcp_nr((1.f + exp_neg_small_(x)) * sig_den_<false>(z, S, mask));
        else {
            t = sig_<FAST>(x);
#pragma unroll
            for (int h = 0; h < 4; ++h) if (h < S) t *= sig_<FAST>(z[h]);
        }
        return log_<FAST>(t + 1e-12f) - log1p_<FAST>(t);
    } else {
        float t = x;
#pragma unroll
        for (int h = 0; h < 4; ++h) if (h < S) t += z[h];
        return log_<FAST>(sig_<FAST>(t) + 1e-12f);
    }
}

template <bool FAST>
__device__ __forceinline__ void fuse2_t(int mode, float x, float m, const float *z, int S, uint32_t mask, float &fx, float &fm) {
    if (mode == 0) {
        if (FAST) {
            const float p = rcp_nr(sig_den_<true>(z, S, mask));
            fx = x * p; fm = m * p;
            return;
        }
        fx = x; fm = m;
#pragma unroll
        for (int h = 0; h < 4; ++h)
            if (h < S && (mask & (1u << h))) { const float sg = sig_<FAST>(z[h]); fx *= sg; fm *= sg; }
    } else if (mode == 1) {
        float tx, tm;
        if (FAST) {
            const float den = sig_den_<false>(z, S, mask);
            tx = rcp_nr((1.f + exp_neg_small_(x)) * den);
            tm = rcp_nr((1.f + exp_neg_small_(m)) * den);
        } else {
            tx = sig_<FAST>(x); tm = sig_<FAST>(m);
#pragma unroll
            for (int h = 0; h < 4; ++h)
                if (h < S) { const float sg = sig_<FAST>(z[h]); tx *= sg; tm *= sg; }
        }
        fx = log_<FAST>(tx + 1e-12f) - log1p_<FAST>(tx);
        fm = log_<FAST>(tm + 1e-12f) - log1p_<FAST>(tm);
    } else {
        float tx = x, tm = m;
#pragma unroll
        for (int h = 0; h < 4; ++h)
            if (h < S) { tx += z[h]; tm += z[h]; }
        fx = log_<FAST>(sig_<FAST>(tx) + 1e-12f);
        fm = log_<FAST>(sig_<FAST>(tm) + 1e-12f);
    }
}
// FAST, rubi fusion, predict type TE / TIE as ONE expression with ONE reciprocal per (user, item) pair: with Q = 1 + e^-a
// (ui = 1 / Q) and D = prod_h (1 + e^-z_h) over the heads of the modality mask,
//   TE  argument  ui / D       = 1 / (Q D)
//   TIE argument  (ui - m) / D = (1 - m Q) / (Q D)            (m = the user's mean of ui over the catalogue)
// a clamped at -80 (Q D stays finite; sigmoid(-80) = 2e-35); zs[h] = z_h * -log2(e), the factor folded into the item norms.
__device__ __forceinline__ float rubi_fast_(float a, const float *zs, int S, uint32_t mask, bool tie, float m) {
    const float q = 1.f + __builtin_amdgcn_exp2f(fmaxf(a, -80.f) * -1.44269502162933349609375f);
    float den = q;
#pragma unroll
    for (int h = 0; h < 4; ++h)
        if (h < S) den *= (mask & (1u << h)) ? 1.f + __builtin_amdgcn_exp2f(zs[h]) : 1.f;      // (a select, not a branch)
    const float p = rcp_nr(den);
    return sig_small_<true>(tie ? fmaf(-m, q, 1.f) * p : p);
}
// the last sigmoid of a score: its argument is bounded for the rubi fusion (a product of sigmoids or a difference of two) and
// for predict type normal (ui itself); the logarithms of hm / sum are not
template <bool FAST> __device__ __forceinline__ float sig_out_(int mode, float x) { return mode == 0 ? sig_small_<FAST>(x) : sig_abs_<FAST>(x); }

struct ScoreArgs {
    const float *Y; int64_t ldy; int64_t U; int64_t I; const int64_t *users; int B; int d; int S;
    uint32_t head_mask; int fusion_mode; int predict_type;
    const float *row_mean;           // [B] mean_i ui (TIE), pass 2 only
    float *partial;                  // pass 1: [n_item_tiles x B] partial row sums of ui
    float *scores; int64_t lds;      // pass 2 output
    const float *sqn;                // [N x (1+S)] squared L2 norms of every head block of every row of Y
    float *tile_max;                 // pass 2, optional: [B x tmax_ld] max score of every 16-item tile (BEFORE masking)
    const float *thr;                // pass 2, chunked top-K (nullable): [B] a lower bound of the user's final K-th best score (from
                                     // the chunks scored so far). A tile's 16 scores of a user are STORED only if their maximum
                                     // reaches it; the tile maxima always are, and the selection never reads an unstored tile
    int64_t tmax_ld;
    int64_t item0, item_end;         // score_t16_kernel: the launch covers items [item0, item_end); scores / tile_max are
                                     // indexed relative to item0 (a chunk of the catalogue when only top-K is wanted)
    const uint4 *planes;             // score_t16b_kernel: the chunk's item rows as three bf16 planes, [item - item0][3][COLS]
    const float *inrm;               // score_t16b_kernel pass 2: 1 / max(|row's head block|, eps) of the chunk's items, [item - item0][S]
    // range invariant: every score of a valid (user, item) pair lies in [lo, hi] -- the image of the last sigmoid over the bounded
    // argument of this (predict type, fusion mode): sigma([0, 1]) for normal / rubi TE, sigma([-1, 1]) for rubi TIE, [0, 1] otherwise;
    // a TIE row mean lies in (0, 1). Checked where it costs nothing: on every RETURNED score (range_check_kernel over the K-lists:
    // a wrongly high score -- round 3's fault, 1.0 on sixteen lanes of one launch in 49 000 -- necessarily enters its user's list)
    // and on every row mean; per score in the scorers' epilogues it cost 6 % of a validation pass (one compare per tile maximum:
    // 1.5 %). Violations add 1 to *range_flag (elimrec_score_range_violations).
    float lo, hi;
    int *range_flag;
};

__device__ __forceinline__ float fuse(int mode, float x, const float *z, int S, uint32_t mask) {
    if (mode == 0) {                 // rubi: EliMRec.py:171-188 (absent modality -> factor 1)
        float r = x;
#pragma unroll
        for (int h = 0; h < kMaxS; ++h) if (h < S && (mask & (1u << h))) r *= sigmoidf_(z[h]);
        return r;
    } else if (mode == 1) {          // hm: :190-199 (every head, regardless of the modality mask)
        float t = sigmoidf_(x);
#pragma unroll
        for (int h = 0; h < kMaxS; ++h) if (h < S) t *= sigmoidf_(z[h]);
        return logf(t + 1e-12f) - log1pf(t);
    } else {                         // sum: :201-210
        float t = x;
#pragma unroll
        for (int h = 0; h < kMaxS; ++h) if (h < S) t += z[h];
        return logf(sigmoidf_(t) + 1e-12f);
    }
}

// fuse(x) and fuse(m) for the same z at once: the per-head sigmoids are computed once and applied to both in the
// order fuse() applies them, so both results have the bits two separate calls give
__device__ __forceinline__ void fuse2(int mode, float x, float m, const float *z, int S, uint32_t mask, float &fx, float &fm) {
    if (mode == 0) {
        fx = x; fm = m;
#pragma unroll
        for (int h = 0; h < kMaxS; ++h)
            if (h < S && (mask & (1u << h))) { const float sg = sigmoidf_(z[h]); fx *= sg; fm *= sg; }
    } else if (mode == 1) {
        float tx = sigmoidf_(x), tm = sigmoidf_(m);
#pragma unroll
        for (int h = 0; h < kMaxS; ++h)
            if (h < S) { const float sg = sigmoidf_(z[h]); tx *= sg; tm *= sg; }
        fx = logf(tx + 1e-12f) - log1pf(tx);
        fm = logf(tm + 1e-12f) - log1pf(tm);
    } else {
        float tx = x, tm = m;
#pragma unroll
        for (int h = 0; h < kMaxS; ++h)
            if (h < S) { tx += z[h]; tm += z[h]; }
        fx = logf(sigmoidf_(tx) + 1e-12f);
        fm = logf(sigmoidf_(tm) + 1e-12f);
    }
}

// MFMA form of the same scorer (used when d % 4 == 0 and the head count fits): a workgroup owns 32 items
// x up to 128 users (the whole evaluation block, so every item row leaves HBM exactly once per block);
// wave w owns users [32w, 32w+32). Per head block the user and item rows are staged 64 columns at a time
// (row stride 65: conflict-free operand reads), their squared norms are accumulated from the same LDS
// image, and v_mfma_f32_32x32x2_f32 accumulates the 32x32 dot products (exact fp32 fma chains).
typedef float v16f_s __attribute__((ext_vector_type(16)));
constexpr int MU = 128, MI = 32, MK = 64, MLD = MK + 1;

template <int PASS>
__global__ __launch_bounds__(256) void score_mfma_kernel(ScoreArgs a) {
    __shared__ float us[MU * MLD];
    __shared__ float it[MI * MLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int64_t i0 = (int64_t)blockIdx.x * MI;
    const int b0 = blockIdx.y * MU;
    const int nblk = (PASS == 1) ? 1 : 1 + a.S;
    v16f_s acc[1 + kMaxS];
#pragma unroll
    for (int h = 0; h < 1 + kMaxS; ++h) acc[h] = (v16f_s){0};
#pragma unroll
    for (int h = 0; h < 1 + kMaxS; ++h) {
        if (h >= nblk) break;
        for (int k0 = 0; k0 < a.d; k0 += MK) {
            const int kc = (a.d - k0) < MK ? (a.d - k0) : MK;
            __syncthreads();
            for (int e = tid * 4; e < MU * kc; e += 1024) {
                const int r = e / kc, c = e - r * kc;
                const int b = b0 + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (b < a.B) v = *reinterpret_cast<const float4 *>(a.Y + a.users[b] * a.ldy + h * a.d + k0 + c);
                float *dst = us + r * MLD + c;
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
            for (int e = tid * 4; e < MI * kc; e += 1024) {
                const int r = e / kc, c = e - r * kc;
                const int64_t item = i0 + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (item < a.I) v = *reinterpret_cast<const float4 *>(a.Y + (a.U + item) * a.ldy + h * a.d + k0 + c);
                float *dst = it + r * MLD + c;
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
            __syncthreads();
            const float *ap = us + (wave * 32 + li) * MLD + lk;
            const float *bp = it + li * MLD + lk;
            for (int k = 0; k < kc; k += 2)
                acc[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k], acc[h], 0, 0, 0);
        }
    }
    const int64_t item = i0 + li;
    const bool item_ok = item < a.I;
    const float eps = 1e-12f;
    const int nb = 1 + a.S;
    float inorm[kMaxS];                               // max(|item block h|, eps), from the precomputed table
#pragma unroll
    for (int h = 0; h < kMaxS; ++h)
        inorm[h] = (PASS == 2 && h < a.S && item_ok) ? fmaxf(sqrtf(a.sqn[(a.U + item) * nb + 1 + h]), eps) : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int urow = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int b = b0 + urow;
        if (PASS == 1) {
            float v = (item_ok && b < a.B) ? sigmoidf_(acc[0][r]) : 0.f;
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);      // over the 32 items of the tile
            if (li == 0 && b < a.B) a.partial[(int64_t)blockIdx.x * a.B + b] = v;
            continue;
        }
        if (b >= a.B || !item_ok) continue;
        const float ui = sigmoidf_(acc[0][r]);
        float out;
        if (a.predict_type == 0) {
            out = sigmoidf_(ui);
        } else {
            float z[kMaxS];
            const int64_t unode = a.users[b];
#pragma unroll
            for (int h = 0; h < kMaxS; ++h)
                z[h] = (h < a.S) ? acc[1 + h][r] / (fmaxf(sqrtf(a.sqn[unode * nb + 1 + h]), eps) * inorm[h]) : 0.f;
            const float te = fuse(a.fusion_mode, ui, z, a.S, a.head_mask);
            if (a.predict_type == 1) out = sigmoidf_(te);
            else out = sigmoidf_(te - fuse(a.fusion_mode, a.row_mean[b], z, a.S, a.head_mask));
        }
        a.scores[(int64_t)b * a.lds + item] = out;
    }
}

// The scorer the evaluator runs (recdim 64, 1-3 single-modal heads): a users-resident 32 x 32 form (removed in round 6) needed 128 registers for a
// wave's 32 users' A operands, 64 accumulators and a 16-output epilogue -- 370 registers, ONE wave per SIMD, nothing to
// hide the epilogue's transcendental chains or the item loads behind (15 % of the fp32 MFMA rate, 18 % MFMA-busy,
// 30 % VALU-busy by the PMC counters). Here a wave owns 16 users (v_mfma_f32_16x16x4_f32: 64 A registers for four
// head blocks, 16 accumulators, 4 outputs per lane), eight waves = the 128 users of an evaluation block share every
// 16-item tile through LDS, and two workgroups fit a CU: four waves per SIMD. The item tile is double-buffered as
// before; its row stride (cols + 4 floats) keeps the b128 stores aligned and the B-operand reads two-way at worst.
constexpr int TU = 16, TI = 16, TW = 8;           // users per wave, items per tile, waves per workgroup
typedef float v4f_s __attribute__((ext_vector_type(4)));

constexpr int t16_sub(int pass) { return pass == 1 ? 1 : 1; }     // 16-item tiles per barrier interval (measured: 4 / 2 are slower -- fewer, fatter workgroups)

template <int PASS, int NB, int PT, int FM, bool FAST, int D>
__global__ __launch_bounds__(512, (D > 64 ? 2 : 4)) void score_t16_kernel(ScoreArgs a, int n_tiles) {
    constexpr int NH = (PASS == 1) ? 1 : NB;
    constexpr int COLS = NH * D, LD = COLS + 4;
    constexpr int SUB = t16_sub(PASS), CI = SUB * TI;  // items per chunk
    extern __shared__ float smem[];
    float *it0 = smem, *it1 = smem + CI * LD;
    float *unorm = it1 + CI * LD;                     // [128][NB-1]
    float *umean = unorm + TW * TU * (NB > 1 ? NB - 1 : 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int b0 = blockIdx.y * (TW * TU);
    const float eps = 1e-12f;
    const int ptype = PT >= 0 ? PT : a.predict_type, fmode = FM >= 0 ? FM : a.fusion_mode;
    const int n_chunks = (n_tiles + SUB - 1) / SUB;
    // A operands: user (wave*16 + li), element k = 4*ks + kq of head block h
    float ua[NH][D / 4];
    {
        const int ub = b0 + wave * TU + li;
        const float *urow = ub < a.B ? a.Y + a.users[ub] * a.ldy + kq : nullptr;
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int ks = 0; ks < D / 4; ++ks) ua[h][ks] = urow ? urow[h * D + 4 * ks] : 0.f;
    }
    if (PASS == 2 && tid < TW * TU) {
        const int b = b0 + tid;
        const int64_t un = b < a.B ? a.users[b] : -1;
        for (int h = 0; h + 1 < NB; ++h) {
            const float nrm = (un >= 0 && ptype != 0) ? fmaxf(sqrtf(a.sqn[un * NB + 1 + h]), eps) : 1.f;
            unorm[tid * (NB - 1) + h] = FAST ? rcp_nr(nrm) : nrm;                      // FAST: reciprocal norms
        }
        umean[tid] = (b < a.B && ptype == 2) ? a.row_mean[b] : 0.f;
    }
    // the chunk's rows as float4 over the 512 threads
    constexpr int PFN = (CI * COLS / 4 + 511) / 512;
    float4 pf[PFN];
    auto load_chunk = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < PFN; ++q) {
            const int e = (tid + 512 * q) * 4;
            if (e < CI * COLS) {
                const int r = e / COLS, c = e - r * COLS;
                const int64_t item = a.item0 + (int64_t)chunk * CI + r;
                pf[q] = item < a.item_end ? *reinterpret_cast<const float4 *>(a.Y + (a.U + item) * a.ldy + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_chunk = [&](float *buf) {
#pragma unroll
        for (int q = 0; q < PFN; ++q) {
            const int e = (tid + 512 * q) * 4;
            if (e < CI * COLS) {
                const int r = e / COLS, c = e - r * COLS;
                *reinterpret_cast<float4 *>(buf + r * LD + c) = pf[q];
            }
        }
    };
    float psum[4] = {0.f, 0.f, 0.f, 0.f};             // PASS 1: this wave's users' running sums of ui, tiles in ascending order
    float thr_r[4];                                   // PASS 2: the store thresholds of this lane's four users
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + wave * TU + 4 * kq + r;
        thr_r[r] = (PASS == 2 && a.thr && b < a.B) ? a.thr[b] : -INFINITY;
    }
    // PASS 2: the squared block norms of this lane's item, fetched one chunk ahead like the rows (a dependent load at the
    // head of every epilogue otherwise)
    constexpr int NQ = (PASS == 2 && NB > 1) ? SUB * (NB - 1) : 1;
    float sq_cur[NQ], sq_nxt[NQ];
    auto load_sqn = [&](int chunk, float (&dst)[NQ]) {
        if (PASS != 2 || NB <= 1) return;
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
            const int64_t item = a.item0 + ((int64_t)chunk * SUB + sub) * TI + li;
#pragma unroll
            for (int h = 0; h + 1 < NB; ++h)
                dst[sub * (NB - 1) + h] = (item < a.item_end && ptype != 0) ? a.sqn[(a.U + item) * NB + 1 + h] : 1.f;
        }
    };
    if ((int)blockIdx.x < n_chunks) { load_chunk(blockIdx.x); load_sqn(blockIdx.x, sq_cur); store_chunk(it0); }
    __syncthreads();
    int cur = 0;
    for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x, cur ^= 1) {
        const int next = chunk + gridDim.x;
        if (next < n_chunks) { load_chunk(next); load_sqn(next, sq_nxt); }
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
            const int tile = chunk * SUB + sub;
            if (tile >= n_tiles) break;               // workgroup-uniform
            const int64_t i0 = a.item0 + (int64_t)tile * TI;
            v4f_s acc[NH];
            const float *bp = (cur ? it1 : it0) + (sub * TI + li) * LD + kq;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                acc[h] = (v4f_s){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < D / 4; ++ks)
                    acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[h][ks], bp[h * D + 4 * ks], acc[h], 0, 0, 0);
            }
            // lane: item i0 + li, users wave*16 + 4*kq + r
            const int64_t item = i0 + li;
            const bool item_ok = item < a.item_end;
            if (PASS == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = b0 + wave * TU + 4 * kq + r;
                    const float v = (item_ok && b < a.B) ? sig_abs_<FAST>(acc[0][r]) : 0.f;
                    psum[r] += v;                     // this lane's item of every tile; the 16 lanes are added once, at the end
                }
            } else {
                float inorm[NB > 1 ? NB - 1 : 1];
#pragma unroll
                for (int h = 0; h + 1 < NB; ++h) {
                    inorm[h] = (item_ok && ptype != 0) ? fmaxf(sqrtf(sq_cur[sub * (NB - 1) + h]), eps) : 1.f;
                    if (FAST) inorm[h] = rcp_nr(inorm[h]);
                }
                const bool rubi_fast = FAST && fmode == 0 && ptype != 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int urow = wave * TU + 4 * kq + r;
                    const float ui = sig_abs_<FAST>(acc[0][r]);
                    float out;
                    if (ptype == 0) {
                        out = sig_small_<FAST>(ui);
                    } else if (rubi_fast) {
                        float zs[kMaxS];
#pragma unroll
                        for (int h = 0; h < kMaxS; ++h)
                            zs[h] = (h + 1 < NB) ? acc[(h + 1 < NH) ? h + 1 : 0][r] * (unorm[urow * (NB - 1) + (h + 1 < NB ? h : 0)] *
                                                                                   (inorm[h + 1 < NB ? h : 0] * -1.44269502162933349609375f)) : 0.f;
                        out = rubi_fast_(acc[0][r], zs, NB - 1, a.head_mask, ptype == 2, umean[urow]);
                    } else {
                        float z[kMaxS];
#pragma unroll
                        for (int h = 0; h < kMaxS; ++h) {
                            const float nn = unorm[urow * (NB - 1) + (h + 1 < NB ? h : 0)] * inorm[h + 1 < NB ? h : 0];
                            const float dp = acc[(h + 1 < NH) ? h + 1 : 0][r];
                            z[h] = (h + 1 < NB) ? (FAST ? dp * nn : dp / nn) : 0.f;
                        }
                        if (ptype == 1) out = sig_out_<FAST>(fmode, fuse_t<FAST>(fmode, ui, z, NB - 1, a.head_mask));
                        else {
                            float te, nde;
                            fuse2_t<FAST>(fmode, ui, umean[urow], z, NB - 1, a.head_mask, te, nde);
                            out = sig_out_<FAST>(fmode, te - nde);
                        }
                    }
                    const bool row_ok = b0 + urow < a.B;
                    bool keep = true;
                    if (a.tile_max) {                               // max over the tile's 16 items (lanes li) of this user
                        const float mx = row16_max_nonneg(item_ok ? out : -INFINITY);
                        if (li == 0 && row_ok) a.tile_max[(int64_t)(b0 + urow) * a.tmax_ld + tile] = mx;
                        keep = !(mx < thr_r[r]);                    // below the user's running K-th best: never read again
                    }
                    if (item_ok && row_ok && keep) a.scores[(int64_t)(b0 + urow) * a.lds + (item - a.item0)] = out;
                }
            }
        }
        if (next < n_chunks) store_chunk(cur ? it0 : it1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) sq_cur[q] = sq_nxt[q];
        __syncthreads();
    }
    if (PASS == 1) {                                  // one partial per (workgroup, user): row_mean_kernel adds them in order
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = b0 + wave * TU + 4 * kq + r;
            const float tot = row16_sum(psum[r]);     // the 16 item lanes of this user, fixed order
            if (li == 0 && b < a.B) a.partial[(int64_t)blockIdx.x * a.B + b] = tot;
        }
    }
}

// ---- pass 2 on the bf16 matrix cores with fp32 results (the FAST math's scorer for recdim 32 / 64)
// An fp32 value splits EXACTLY into three bf16 pieces x = x1 + x2 + x3 (8 + 8 + 8 significand bits, by truncation), a product
// of two pieces is exact in fp32, and of the nine piece products of a.b the six with i + j <= 4 carry everything above
// 2^-24 relative: a.b = a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1) to fp32 round-off, accumulated in the MFMA's fp32
// accumulator. v_mfma_f32_16x16x32_bf16 retires 8192 MACs in 16 cycles against 1024 in 32 for v_mfma_f32_16x16x4_f32: six
// of the former replace sixteen of the latter per 16 x 16 x 64 block -- 2.7x less matrix-core time for the same scores.
// The items' pieces are made once per catalogue chunk (split3_items_kernel, 16 MB read + 25 MB written against a 0.9 ms
// scorer launch), the users' pieces once per workgroup, in registers.
typedef __bf16 bf16x8_s __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, uint32_t &p1, uint32_t &p2, uint32_t &p3) {      // bf16 bit patterns in the HIGH half
    p1 = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(p1);
    p2 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(p2);
    p3 = __float_as_uint(r2) & 0xffff0000u;
}

// 8 consecutive floats -> three uint4 of 8 bf16 each (element t in bits [16 (t & 1), +16) of word t >> 1)
__device__ __forceinline__ void split3x8(const float4 &lo, const float4 &hi, uint4 &q1, uint4 &q2, uint4 &q3) {
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t w1[4], w2[4], w3[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        uint32_t a1, a2, a3, b1, b2, b3;
        split3(x[2 * t], a1, a2, a3);
        split3(x[2 * t + 1], b1, b2, b3);
        w1[t] = (a1 >> 16) | b1; w2[t] = (a2 >> 16) | b2; w3[t] = (a3 >> 16) | b3;
    }
    q1 = make_uint4(w1[0], w1[1], w1[2], w1[3]); q2 = make_uint4(w2[0], w2[1], w2[2], w2[3]); q3 = make_uint4(w3[0], w3[1], w3[2], w3[3]);
}

// planes[(item - item0)][p][c] (bf16) <- piece p of Y[U + item][c], c < cols (a multiple of 8): a thread per 8 columns
// inrm (pass 2, nullable): the items' inverse head-block norms, once per chunk instead of once per (user group, tile) -- the
// expression the scorers evaluate: rcp_nr(max(sqrtf(sqn), eps)); nb = 1 + heads, sqn [N x nb]
__global__ __launch_bounds__(256) void split3_items_kernel(const float *__restrict__ Y, int64_t ldy, int64_t U, int64_t item0,
                                                           int64_t n_items, int cols, uint4 *__restrict__ planes,
                                                           const float *__restrict__ sqn, int nb, float *__restrict__ inrm) {
    const int c8 = cols / 8;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_items * c8) return;
    const int64_t it = t / c8;
    const int c = (int)(t % c8);
    if (inrm && c + 1 < nb) inrm[it * (nb - 1) + c] = rcp_nr(fmaxf(sqrtf(sqn[(U + item0 + it) * nb + 1 + c]), 1e-12f));
    const float4 *src = reinterpret_cast<const float4 *>(Y + (U + item0 + it) * ldy + 8 * c);
    uint4 q1, q2, q3;
    split3x8(src[0], src[1], q1, q2, q3);
    uint4 *dst = planes + it * 3 * c8 + c;
    dst[0] = q1; dst[c8] = q2; dst[2 * c8] = q3;
}

// (Measured: four waves / 64 users per workgroup and two workgroups per CU -- so that the per-tile barrier ties four waves
// together and the CU's two workgroups drift apart -- 0.0169 s per validation pass against 0.0157 s: the item tile is then
// staged twice per CU.)
constexpr int TWB = 8, NTB = 64 * TWB;
// tiles per barrier window of score_t16b_kernel (ring of 2 * window item-tile buffers in LDS): as many as 150 KB hold, at most 4
constexpr int t16b_row4(int pass, int nb, int d) { return 3 * ((pass == 1 ? 1 : nb) * d) / 8 + 1; }
constexpr int t16b_win(int pass, int nb, int d) {
    int w = 4;
    while (w > 1 && (size_t)2 * w * 16 * t16b_row4(pass, nb, d) * 16 > (size_t)150 * 1024) --w;
    return w;
}
template <int PASS, int NB, int PT, int FM, int D>
__global__ __launch_bounds__(NTB, 2) void score_t16b_kernel(ScoreArgs a, int n_tiles) {
    constexpr bool FAST = true;
    constexpr int NH = (PASS == 1) ? 1 : NB, COLS = NH * D;
    constexpr int ROW4 = 3 * COLS / 8 + 1;            // LDS row of an item in uint4 units: three planes + 16 B of padding
    constexpr int KB = D / 32;                         // 32-deep MFMA steps per head block
    extern __shared__ float smem[];
    // 2 * WIN item-tile buffers in a ring, tiles staged WIN ahead, ONE barrier per WIN tiles: inside a window the waves read
    // buffers k .. k + WIN - 1 and write k + WIN .. k + 2 WIN - 1 (mod 2 WIN) -- the ones the window before read, which every wave
    // has left behind at the barrier in between -- so the eight waves drift up to WIN - 1 tiles apart and one SIMD's two waves
    // overlap their MFMA and epilogue phases instead of meeting at a barrier after every 16 items (WIN = 3 at recdim 64 with
    // three heads: 149 KB of LDS; one workgroup per CU either way -- the kernel's 218 registers allow two waves per SIMD)
    constexpr int WIN = t16b_win(PASS, NB, D), RING = 2 * WIN;
    static_assert(ROW4 == t16b_row4(PASS, NB, D), "LDS row size");
    uint4 *itb = reinterpret_cast<uint4 *>(smem);
    float *unorm = reinterpret_cast<float *>(itb + RING * TI * ROW4);     // [128][NB-1]
    float *umean = unorm + TWB * TU * (NB > 1 ? NB - 1 : 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int b0 = blockIdx.y * (TWB * TU);
    const float eps = 1e-12f;
    const int ptype = PT >= 0 ? PT : a.predict_type, fmode = FM >= 0 ? FM : a.fusion_mode;
    // A operands: user (wave*16 + li), pieces of elements k = 32 j + 8 kq .. + 8 of head block h
    uint4 ua[3][NH][KB];
    {
        const int ub = b0 + wave * TU + li;
        const float *urow = ub < a.B ? a.Y + a.users[ub] * a.ldy : nullptr;
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
                if (urow) {
                    const float4 *s4 = reinterpret_cast<const float4 *>(urow + h * D + 32 * j + 8 * kq);
                    lo = s4[0]; hi = s4[1];
                }
                split3x8(lo, hi, ua[0][h][j], ua[1][h][j], ua[2][h][j]);
            }
    }
    if (PASS == 2 && tid < TWB * TU) {
        const int b = b0 + tid;
        const int64_t un = b < a.B ? a.users[b] : -1;
        for (int h = 0; h + 1 < NB; ++h) {
            const float nrm = (un >= 0 && ptype != 0) ? fmaxf(sqrtf(a.sqn[un * NB + 1 + h]), eps) : 1.f;
            unorm[tid * (NB - 1) + h] = rcp_nr(nrm);
        }
        umean[tid] = (b < a.B && ptype == 2) ? a.row_mean[b] : 0.f;
    }
    float psum[4] = {0.f, 0.f, 0.f, 0.f};             // PASS 1: this wave's users' running sums of ui, tiles in ascending order
    float thr_r[4];                                   // PASS 2: the store thresholds of this lane's four users
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + wave * TU + 4 * kq + r;
        thr_r[r] = (PASS == 2 && a.thr && b < a.B) ? a.thr[b] : -INFINITY;
    }
    // a tile's rows (16 items x three planes) as uint4 over the workgroup's threads
    constexpr int TILE4 = TI * 3 * COLS / 8;
    constexpr int PFN = (TILE4 + NTB - 1) / NTB;
    uint4 pf[PFN];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int q = 0; q < PFN; ++q) {
            const int e = tid + NTB * q;
            if (e < TILE4) {
                const int64_t it = (int64_t)tile * TI + e / (3 * COLS / 8);
                pf[q] = (a.item0 + it < a.item_end) ? a.planes[it * (3 * COLS / 8) + e % (3 * COLS / 8)] : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    auto store_tile = [&](uint4 *buf) {
#pragma unroll
        for (int q = 0; q < PFN; ++q) {
            const int e = tid + NTB * q;
            if (e < TILE4) buf[(e / (3 * COLS / 8)) * ROW4 + e % (3 * COLS / 8)] = pf[q];
        }
    };
    constexpr int NQ = (PASS == 2 && NB > 1) ? NB - 1 : 1;
    float sq_cur[NQ], sq_nxt[NQ];
    auto load_sqn = [&](int tile, float (&dst)[NQ]) {       // (the INVERSE norms of the tile's items: split3_items_kernel made them)
        if (PASS != 2 || NB <= 1) return;
        const int64_t it = (int64_t)tile * TI + li;
#pragma unroll
        for (int h = 0; h + 1 < NB; ++h) dst[h] = (a.item0 + it < a.item_end && ptype != 0) ? a.inrm[it * (NB - 1) + h] : 1.f;
    };
    // a workgroup takes a CONTIGUOUS stretch of the launch's tiles: the maxima of 16 consecutive tiles of a user then leave as one
    // 64-B store (below) instead of sixteen 4-B ones, each of which cost a 32-B write at the memory side
    const int per = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = (int)blockIdx.x * per, t_end = t_begin + per < n_tiles ? t_begin + per : n_tiles;
    if (t_begin < t_end) load_sqn(t_begin, sq_cur);
#pragma unroll
    for (int w = 0; w < WIN; ++w)
        if (t_begin + w < t_end) { load_tile(t_begin + w); store_tile(itb + w * (TI * ROW4)); }
    __syncthreads();
    float mxb[4] = {0.f, 0.f, 0.f, 0.f};              // PASS 2: lane li holds the maximum of tile (base + li) of its four users
    int ring = 0, in_win = 0;                         // this tile's buffer; its place in the barrier window
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int next = tile + 1, ahead = tile + WIN;
        if (ahead < t_end) load_tile(ahead);
        if (next < t_end) load_sqn(next, sq_nxt);
        const int64_t i0 = a.item0 + (int64_t)tile * TI;
        v4f_s acc[NH];
        const uint4 *brow = itb + ring * (TI * ROW4) + li * ROW4 + kq;   // this lane's item row, its k-block of 8
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            acc[h] = (v4f_s){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                const int c8 = (h * D + 32 * j) / 8;
                const bf16x8_s b1 = __builtin_bit_cast(bf16x8_s, brow[c8]);
                const bf16x8_s b2 = __builtin_bit_cast(bf16x8_s, brow[COLS / 8 + c8]);
                const bf16x8_s b3 = __builtin_bit_cast(bf16x8_s, brow[2 * (COLS / 8) + c8]);
                const bf16x8_s a1 = __builtin_bit_cast(bf16x8_s, ua[0][h][j]), a2 = __builtin_bit_cast(bf16x8_s, ua[1][h][j]),
                               a3 = __builtin_bit_cast(bf16x8_s, ua[2][h][j]);
                // smallest terms first
                acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, acc[h], 0, 0, 0);
                acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, acc[h], 0, 0, 0);
                acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, acc[h], 0, 0, 0);
                acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc[h], 0, 0, 0);
                acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, acc[h], 0, 0, 0);
                acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[h], 0, 0, 0);
            }
        }
        // lane: item i0 + li, users wave*16 + 4*kq + r  (the 16x16 output layout of the fp32 form)
        const int64_t item = i0 + li;
        const bool item_ok = item < a.item_end;
        float inorm[NQ];
        const bool rubi_fast = PASS == 2 && fmode == 0 && ptype != 0;
        if (PASS == 2) {
#pragma unroll
            for (int h = 0; h + 1 < NB; ++h) {
                inorm[h] = sq_cur[h];                          // rcp_nr(max(sqrtf(sqn), eps)), or 1 beyond the chunk / for type normal
                if (rubi_fast) inorm[h] *= -1.44269502162933349609375f;      // rubi_fast_ takes z_h * -log2(e)
            }
        }
        // all four users' scores first -- four independent chains of transcendentals in ONE basic block, so that the scheduler can
        // interleave them (a predicated store after each would end the block) -- then the stores and the tile maxima
        float outs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int urow = wave * TU + 4 * kq + r;
            if (PASS == 1) {
                const float ui = sig_mean_<FAST>(acc[0][r]);
                psum[r] += (item_ok && b0 + urow < a.B) ? ui : 0.f;      // this lane's item of every tile; the 16 lanes are added at the end
                continue;
            }
            float out;
            if (ptype == 0) {
                out = sig_small_<FAST>(sig_abs_<FAST>(acc[0][r]));
            } else if (rubi_fast) {
                float zs[kMaxS];
#pragma unroll
                for (int h = 0; h < kMaxS; ++h)
                    zs[h] = (h + 1 < NB) ? acc[(h + 1 < NH) ? h + 1 : 0][r] * (unorm[urow * (NB - 1) + (h + 1 < NB ? h : 0)] * inorm[h + 1 < NB ? h : 0]) : 0.f;
                out = rubi_fast_(acc[0][r], zs, NB - 1, a.head_mask, ptype == 2, umean[urow]);
            } else {
                const float ui = sig_abs_<FAST>(acc[0][r]);
                float z[kMaxS];
#pragma unroll
                for (int h = 0; h < kMaxS; ++h) {
                    const float nn = unorm[urow * (NB - 1) + (h + 1 < NB ? h : 0)] * inorm[h + 1 < NB ? h : 0];
                    const float dp = acc[(h + 1 < NH) ? h + 1 : 0][r];
                    z[h] = (h + 1 < NB) ? dp * nn : 0.f;
                }
                if (ptype == 1) out = sig_out_<FAST>(fmode, fuse_t<FAST>(fmode, ui, z, NB - 1, a.head_mask));
                else {
                    float te, nde;
                    fuse2_t<FAST>(fmode, ui, umean[urow], z, NB - 1, a.head_mask, te, nde);
                    out = sig_out_<FAST>(fmode, te - nde);
                }
            }
            outs[r] = out;
        }
        if (PASS == 2) {
            // the tile maxima first (always stored), then the scores -- of the users whose maximum reaches their threshold only:
            // a wave none of whose 16 users keeps the tile (the usual case once a chunk or two have been scored) skips the stores
            bool keep[4];
            bool any = !a.tile_max;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int urow = wave * TU + 4 * kq + r;
                const bool row_ok = b0 + urow < a.B;
                keep[r] = true;
                if (a.tile_max) {
                    const float mx = row16_max_nonneg(item_ok ? outs[r] : -INFINITY);
                    // (the four chains above are complete: predicated stores no longer cut the block they are scheduled in)
                    const int slot = (tile - t_begin) & 15;
                    if (li == slot) mxb[r] = mx;
                    if (slot == 15 || tile == t_end - 1) {           // 16 maxima of this user (fewer at the stretch's end): one 64-B store
                        if (row_ok && li <= slot) a.tile_max[(int64_t)(b0 + urow) * a.tmax_ld + (tile - slot) + li] = mxb[r];
                    }
                    keep[r] = !(mx < thr_r[r]);
                    any = any || keep[r];
                }
            }
            if (__builtin_amdgcn_ballot_w64(any) != 0ull) {       // wave-uniform
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int urow = wave * TU + 4 * kq + r;
                    const bool row_ok = b0 + urow < a.B;
                    if (item_ok && row_ok && keep[r]) a.scores[(int64_t)(b0 + urow) * a.lds + (item - a.item0)] = outs[r];
                }
            }
        }
        if (ahead < t_end) store_tile(itb + (ring + WIN >= RING ? ring + WIN - RING : ring + WIN) * (TI * ROW4));
#pragma unroll
        for (int q = 0; q < NQ; ++q) sq_cur[q] = sq_nxt[q];
        ring = ring + 1 == RING ? 0 : ring + 1;
        if (++in_win == WIN) { in_win = 0; __syncthreads(); }      // (workgroup-uniform) the end of a window
    }
    if (PASS == 1) {                                  // one partial per (workgroup, user): row_mean_kernel adds them in order
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = b0 + wave * TU + 4 * kq + r;
            const float tot = row16_sum(psum[r]);     // the 16 item lanes of this user, fixed order
            if (li == 0 && b < a.B) a.partial[(int64_t)blockIdx.x * a.B + b] = tot;
        }
    }
}

// bitmap of the items to mask, one workgroup per user row: clear, then set (rows are disjoint, the atomics stay in one row)
__global__ __launch_bounds__(256) void train_bits_kernel(const int64_t *__restrict__ ptr, const int32_t *__restrict__ items, int64_t I,
                                                         uint32_t *__restrict__ bits, int64_t bits_ld) {
    const int b = blockIdx.x;
    uint32_t *row = bits + (int64_t)b * bits_ld;
    for (int64_t w = threadIdx.x; w < bits_ld; w += 256) row[w] = 0u;
    __syncthreads();
    for (int64_t j = ptr[b] + threadIdx.x; j < ptr[b + 1]; j += 256) {
        const int it = items[j];
        if (it >= 0 && it < I) atomicOr(&row[it >> 5], 1u << (it & 31));
    }
}

// sqn[row, h] = sum_k Y[row, h*d + k]^2 : one wave per row, lanes stride the block, fixed-order reduction.
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float *__restrict__ Y, int64_t ldy, int64_t n_rows, int d,
                                                         int nb, float *__restrict__ sqn) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    for (int h = 0; h < nb; ++h) {
        float s = 0.f;
        for (int k = lane * 4; k < d; k += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(Y + row * ldy + h * d + k);
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        s = wave_sum(s);
        if (lane == 0) sqn[row * nb + h] = s;
    }
}

// mean_i ui for one user per workgroup: thread t adds tiles t, t+256, ... then a fixed binary tree.
__global__ __launch_bounds__(256) void row_mean_kernel(const float *__restrict__ partial, int n_tiles, int B, int64_t I,
                                                       float *__restrict__ row_mean) {
    __shared__ float sm[256];
    const int b = blockIdx.x;
    float s = 0.f;
    for (int t = threadIdx.x; t < n_tiles; t += 256) s += partial[(int64_t)t * B + b];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) row_mean[b] = sm[0] / (float)I;
}

// item-sharded evaluation: the row sums of sigmoid(u.i) were added over the shards by the caller (all_reduce)
__global__ void mean_from_sum_kernel(const float *__restrict__ row_sum, int B, int64_t I_total, float *__restrict__ row_mean) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) row_mean[b] = row_sum[b] / (float)I_total;
}

// The range invariant on what a call RETURNS: every listed score (ids >= 0, not -inf: a masked filler of the reference's order)
// inside [lo, hi], every TIE row mean inside (0, 1). One count per offending user row / mean into *flag.
__global__ __launch_bounds__(256) void range_check_kernel(const float *__restrict__ vals, const int32_t *__restrict__ idx, int B, int K,
                                                          float lo, float hi, const float *__restrict__ mean, int *__restrict__ flag) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    bool bad = false;
    if (vals)
        for (int k = 0; k < K; ++k) {
            const float v = vals[(int64_t)b * K + k];
            if (idx[(int64_t)b * K + k] >= 0 && v != -INFINITY) bad = bad || !(v >= lo && v <= hi);
        }
    if (mean) bad = bad || !(mean[b] > 0.f && mean[b] < 1.f);
    if (bad) atomicAdd(flag, 1);
}

// ... and this shard's item ids become catalogue ids
__global__ void add_id_offset_kernel(int32_t *__restrict__ idx, int64_t n, int32_t off) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && idx[i] >= 0) idx[i] += off;
}

__global__ void mask_train_kernel(float *__restrict__ scores, int64_t lds, const int64_t *__restrict__ ptr,
                                  const int32_t *__restrict__ items, int B) {
    const int b = blockIdx.x;
    if (b >= B) return;
    for (int64_t j = ptr[b] + threadIdx.x; j < ptr[b + 1]; j += blockDim.x)
        scores[(int64_t)b * lds + items[j]] = -INFINITY;
}

// Top-K by (score desc, index asc): K rounds; round r takes the best element strictly after the
// previous pick in that total order. One workgroup per row.
// K rounds over one row by the whole workgroup (any multiple of 64 threads <= 1024): round r takes the best element strictly
// after the previous pick in the order (score desc, index asc). item0: the row holds items [item0, item0 + I) of the catalogue
// (a chunk when only top-K is wanted); results go to oi[rank] / ov[rank] (global or LDS) with catalogue item ids, -1 / -inf
// where the row runs out. tmx / floor_v (nullable): only elements of 16-item tiles whose maximum tmx[i / 16] reaches floor_v
// were stored by the scorer -- the others are not read (they cannot reach the final top-K).
__device__ void topk_rounds(const float *__restrict__ row, int64_t I, int K, int b, int32_t *oi, float *ov,
                            const uint32_t *__restrict__ mask_bits, int64_t bits_ld, int64_t item0,
                            const float *__restrict__ tmx = nullptr, float floor_v = -INFINITY) {
    __shared__ float sv[16];
    __shared__ int si[16];
    __shared__ float last_v;
    __shared__ int last_i;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float pv = INFINITY;
    int pi = -1;
    for (int r = 0; r < K; ++r) {
        float bv = -INFINITY;
        int bi = INT32_MAX;
        for (int64_t i = threadIdx.x; i < I; i += blockDim.x) {
            if (tmx && tmx[i / TI] < floor_v) continue;                       // an unstored tile
            float v = row[i];
            if (mask_bits && ((mask_bits[(int64_t)b * bits_ld + ((item0 + i) >> 5)] >> ((item0 + i) & 31)) & 1u)) v = -INFINITY;   // a masked item
            const bool after = (r == 0) || (v < pv) || (v == pv && (int)i > pi);
            if (after && (v > bv || (v == bv && (int)i < bi))) { bv = v; bi = (int)i; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov2 = __shfl_xor(bv, off, 64);
            const int oi2 = __shfl_xor(bi, off, 64);
            if (ov2 > bv || (ov2 == bv && oi2 < bi)) { bv = ov2; bi = oi2; }
        }
        if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float fv = sv[0];
            int fi = si[0];
            for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
                if (sv[w] > fv || (sv[w] == fv && si[w] < fi)) { fv = sv[w]; fi = si[w]; }
            last_v = fv; last_i = fi;
            oi[r] = (fi == INT32_MAX) ? -1 : (int32_t)(item0 + fi);
            if (ov) ov[r] = fv;
        }
        __syncthreads();
        pv = last_v; pi = last_i;
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void topk_kernel(const float *__restrict__ scores, int64_t lds, int64_t I, int K,
                                                    int32_t *__restrict__ out_idx, float *__restrict__ out_val,
                                                    const int32_t *__restrict__ only_if,
                                                    const uint32_t *__restrict__ mask_bits, int64_t bits_ld,
                                                    int64_t item0, int64_t ldo, int64_t oo) {
    const int b = blockIdx.x;
    if (only_if && only_if[b] == 0) return;        // this row was finished by topk_select_kernel
    topk_rounds(scores + (int64_t)b * lds, I, K, b, out_idx + (int64_t)b * ldo + oo, out_val ? out_val + (int64_t)b * ldo + oo : nullptr,
                mask_bits, bits_ld, item0);
}

// Fast path of the same selection (identical result): two sweeps of the row instead of K.
//   1. every thread keeps the max of its strided elements; G >= 2K group maxima are formed and the K-th
//      largest of them, tau, is found by rank counting -- at least K elements are >= tau, so the top-K is
//      among the elements >= tau (about 1.6 K of them for continuous scores);
//   2. those candidates are compacted into LDS and ranked by (score desc, index asc) by counting, entry
//      with rank r < K is output r.
// If more than TK_CAP candidates qualify (massive ties, or fewer than K finite scores) the workgroup falls
// back to the K-round sweep below, so the result never depends on which path ran.
constexpr int TK_CAP = 1024;
constexpr int RM_KMAX = 256;          // largest K of the tile-guided selection and of the metric kernel
__device__ __forceinline__ bool tk_before(float va, int ia, float vb, int ib) {   // a ranks before b
    return va > vb || (va == vb && ia < ib);
}

__global__ __launch_bounds__(1024) void topk_select_kernel(const float *__restrict__ scores, int64_t lds, int64_t I,
                                                           int K, int G, int32_t *__restrict__ out_idx,
                                                           float *__restrict__ out_val, int32_t *__restrict__ fallback) {
    __shared__ float tm[1024];
    __shared__ float gm[1024];
    __shared__ float cv[TK_CAP];
    __shared__ int ci[TK_CAP];
    __shared__ float tau;
    __shared__ int cnt;
    const int b = blockIdx.x, t = threadIdx.x;
    const float *row = scores + (int64_t)b * lds;
    float m = -INFINITY;
    for (int64_t i = t; i < I; i += 1024) m = fmaxf(m, row[i]);
    tm[t] = m;
    if (t == 0) cnt = 0;
    __syncthreads();
    const int gs = 1024 / G;                       // threads per group
    if (t < G) {
        float g = -INFINITY;
        for (int j = 0; j < gs; ++j) g = fmaxf(g, tm[t * gs + j]);
        gm[t] = g;
    }
    __syncthreads();
    if (t < G) {                                   // rank of gm[t] among the G group maxima (ties by index)
        int rank = 0;
        const float mine = gm[t];
        for (int j = 0; j < G; ++j) rank += tk_before(gm[j], j, mine, t) ? 1 : 0;
        if (rank == K - 1) tau = mine;
    }
    __syncthreads();
    const float th = tau;
    for (int64_t i = t; i < I; i += 1024) {
        const float v = row[i];
        if (v >= th) {
            const int slot = atomicAdd(&cnt, 1);
            if (slot < TK_CAP) { cv[slot] = v; ci[slot] = (int)i; }
        }
    }
    __syncthreads();
    const int n = cnt;
    if (n > TK_CAP || n < K || th == -INFINITY) {  // workgroup-uniform
        if (t == 0) fallback[b] = 1;
        return;
    }
    if (t == 0) fallback[b] = 0;
    if (t < n) {
        int rank = 0;
        const float mv = cv[t];
        const int mi = ci[t];
        for (int j = 0; j < n; ++j) rank += tk_before(cv[j], ci[j], mv, mi) ? 1 : 0;
        if (rank < K) {
            out_idx[(int64_t)b * K + rank] = mi;
            if (out_val) out_val[(int64_t)b * K + rank] = mv;
        }
    }
}

// The same selection guided by the scorer's tile maxima: the row is never swept. Thread t folds the maxima of tiles t,
// t+256, ... into a group maximum; tau = the (K + n_masked)-th largest of the 256 group maxima. The maxima are taken
// BEFORE the train items are masked (a mask test per score in the scorer's epilogue costs more than the whole selection):
// at most n_masked groups owe their maximum to a masked item, so at least K groups -- K unmasked scores -- are >= tau,
// and all of them sit in tiles whose maximum is >= tau, a few dozen of the 4 756 at the Tiktok shape. Only those tiles'
// scores are read, the masked ones dropped (bitmap), the rest compacted and ranked as above. Same fall-back protocol
// (K + n_masked > 256 groups, massive ties, fewer than K unmasked scores).
__global__ __launch_bounds__(256) void topk_tiles_kernel(const float *__restrict__ scores, int64_t lds, int64_t I,
                                                         const float *__restrict__ tile_max, int64_t tmax_ld, int n_tiles, int K,
                                                         const int64_t *__restrict__ mask_ptr, const uint32_t *__restrict__ mask_bits,
                                                         int64_t bits_ld, int32_t *out_idx, float *out_val,
                                                         int32_t *__restrict__ fallback, int64_t item0, int64_t ldo, int64_t oo,
                                                         float *thr, int first_chunk) {
    // thr (nullable) = the RUNNING form of a catalogue scored chunk by chunk: out_idx / out_val [b][K] hold the user's best K of the
    // chunks selected so far (-1 ids where there are fewer) and thr[b] their K-th score (-inf below K entries). The scorer stored
    // this chunk's scores only in tiles whose maximum reaches thr[b]; the selection takes the chunk's candidates >= max(tau, thr[b])
    // -- possibly fewer than K, possibly none --, ranks them TOGETHER with the running list by the same (score desc, id asc)
    // counting and leaves the new list and threshold: after the last chunk the list is the catalogue's top-K, the list the
    // whole-catalogue selection gives (top-K of a union = top-K of the parts' top-Ks under one total order).
    constexpr int POOL = TK_CAP + RM_KMAX;
    __shared__ float gm[256];
    __shared__ float cv[POOL];
    __shared__ int ci[POOL];
    __shared__ int ct[TK_CAP];
    __shared__ float tau;
    __shared__ int cnt, n_ct, n_run;
    const int b = blockIdx.x, t = threadIdx.x;
    const float *row = scores + (int64_t)b * lds;
    const float *tmx = tile_max + (int64_t)b * tmax_ld;
    const bool running = thr != nullptr;
    const float floor_v = (running && !first_chunk) ? thr[b] : -INFINITY;       // (first chunk: no list yet, whatever the buffers hold)
    int32_t *oi = out_idx + (int64_t)b * ldo + oo;
    float *ov = out_val ? out_val + (int64_t)b * ldo + oo : nullptr;
    // the bitmap words of this row's item range (item0 is a multiple of 32)
    const uint32_t *bits = mask_bits ? mask_bits + (int64_t)b * bits_ld + (item0 >> 5) : nullptr;
    __shared__ int n_msk;
    float m = -INFINITY;
    for (int i = t; i < n_tiles; i += 256) m = fmaxf(m, tmx[i]);
    gm[t] = m;
    if (t == 0) { cnt = 0; n_ct = 0; tau = -INFINITY; n_msk = 0; n_run = 0; }
    __syncthreads();
    if (bits && mask_ptr) {          // masked items inside the range: the whole list when the range is the catalogue
        if (item0 == 0 && I + 31 >= bits_ld * 32) { if (t == 0) n_msk = (int)(mask_ptr[b + 1] - mask_ptr[b]); }
        else {
            int c = 0;
            for (int64_t w = t; w < (I + 31) / 32; w += 256) c += __popc(bits[w]);
            if (c) atomicAdd(&n_msk, c);
        }
    }
    __syncthreads();
    const int64_t n_masked = n_msk;
    // tau = the (K + n_masked)-th largest of a set of GROUP maxima: that many different scores are >= tau, so at least K unmasked
    // ones are. 64 coarse groups ranked by one wave when K + n_masked fits (the usual case: 4 096 comparisons instead of 65 536
    // -- this ranking was 85 % of the kernel's instructions; the coarser threshold admits one or two candidates more), the 256
    // fine groups otherwise.
    if (K - 1 + n_masked < 64) {
        __shared__ float g64[64];
        if (t < 64) g64[t] = fmaxf(fmaxf(gm[4 * t], gm[4 * t + 1]), fmaxf(gm[4 * t + 2], gm[4 * t + 3]));
        __syncthreads();
        if (t < 64) {
            const float mine = g64[t];
            int rank = 0;
            for (int j = 0; j < 64; ++j) rank += tk_before(g64[j], j, mine, t) ? 1 : 0;
            if ((int64_t)rank == K - 1 + n_masked) tau = mine;
        }
    } else {
        int rank = 0;
        for (int j = 0; j < 256; ++j) rank += tk_before(gm[j], j, m, t) ? 1 : 0;
        if ((int64_t)rank == K - 1 + n_masked) tau = m;
    }
    __syncthreads();
    const float tau_c = tau;
    const float th = fmaxf(tau_c, floor_v);
    if (th != -INFINITY) {
        // candidate tiles first (their numbers into ct[]), then one score per thread
        for (int i = t; i < n_tiles; i += 256) {
            if (tmx[i] >= th) {
                const int slot = atomicAdd(&n_ct, 1);
                if (slot < TK_CAP) ct[slot] = i;
            }
        }
        __syncthreads();
        const int nct = n_ct < TK_CAP ? n_ct : TK_CAP;
        if (n_ct > TK_CAP && t == 0) cnt = TK_CAP + 1;                 // too many tiles: fall back
        for (int w = t; w < nct * TI; w += 256) {
            const int64_t i = (int64_t)ct[w / TI] * TI + (w % TI);
            if (i >= I) continue;
            const float v = row[i];
            if (v >= th && !(bits && ((bits[i >> 5] >> (i & 31)) & 1u))) {
                const int slot = atomicAdd(&cnt, 1);
                if (slot < TK_CAP) { cv[slot] = v; ci[slot] = (int)(item0 + i); }
            }
        }
    }
    __syncthreads();
    int n = cnt;
    // the chunk must bring K candidates unless the running threshold (not the chunk's own tau) is what cut them
    const bool short_ok = running && floor_v > tau_c;
    if (n > TK_CAP || (n < K && !short_ok) || th == -INFINITY) {   // workgroup-uniform: the K-round sweep, by this workgroup (rare)
        if (t == 0 && fallback) fallback[b] = 1;
        if (!running) {
            topk_rounds(row, I, K, b, oi, ov, mask_bits, bits_ld, item0);
            return;
        }
        // (over the stored tiles only; its picks become the chunk's candidates)
        topk_rounds(row, I, K, b, ci, cv, mask_bits, bits_ld, item0, tmx, floor_v);
        __syncthreads();
        if (t == 0) {                                                  // compact: the sweep leaves -1 ids where the row ran out
            int k = 0;
            for (int j = 0; j < K; ++j)
                if (ci[j] >= 0 && cv[j] != -INFINITY) { ci[k] = ci[j]; cv[k] = cv[j]; ++k; }
            cnt = k;
        }
        __syncthreads();
        n = cnt;
    } else if (t == 0 && fallback) fallback[b] = 0;
    if (running) {                       // the running list joins the pool (read before anything is written back)
        for (int r = t; r < K; r += 256) {
            const int id = first_chunk ? -1 : oi[r];
            if (id >= 0) {
                const int slot = n + atomicAdd(&n_run, 1);
                cv[slot] = ov[r]; ci[slot] = id;
            }
        }
        __syncthreads();
        n += n_run;
        __syncthreads();
        for (int r = t; r < K; r += 256)
            if (r >= n) { oi[r] = -1; ov[r] = -INFINITY; }
        if (t == 0 && n < K) thr[b] = -INFINITY;
    }
    for (int c = t; c < n; c += 256) {
        int rank = 0;
        const float mv = cv[c];
        const int mi = ci[c];
        for (int j = 0; j < n; ++j) rank += tk_before(cv[j], ci[j], mv, mi) ? 1 : 0;
        if (rank < K) {
            oi[rank] = (int32_t)mi;
            if (ov) ov[rank] = mv;
            if (running && rank == K - 1) thr[b] = mv;
        }
    }
}

// Top-K of the per-chunk top-K lists (catalogue scored chunk by chunk: no [B x I] score matrix when only top-K is wanted):
// cand_* [B x n] hold n = chunks * K (score, item id) pairs per user, -1 ids for the slots a chunk could not fill. Ranked by
// (score desc, id asc) by counting -- the order of the whole-catalogue selection, so both give the same list.
__global__ __launch_bounds__(256) void topk_merge_kernel(const float *__restrict__ cand_val, const int32_t *__restrict__ cand_idx,
                                                         int n, int K, int32_t *__restrict__ out_idx, float *__restrict__ out_val) {
    extern __shared__ float mrg[];                   // [n] values, [n] ids
    float *cv = mrg;
    int *ci = (int *)(mrg + n);
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < n; i += 256) { cv[i] = cand_val[(int64_t)b * n + i]; ci[i] = cand_idx[(int64_t)b * n + i]; }
    for (int r = threadIdx.x; r < K; r += 256) { out_idx[(int64_t)b * K + r] = -1; if (out_val) out_val[(int64_t)b * K + r] = -INFINITY; }
    __syncthreads();
    for (int c = threadIdx.x; c < n; c += 256) {
        const float mv = cv[c];
        const int mi = ci[c];
        if (mi < 0) continue;
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (ci[j] >= 0 && tk_before(cv[j], ci[j], mv, mi)) ? 1 : 0;
        if (rank < K) {
            out_idx[(int64_t)b * K + rank] = mi;
            if (out_val) out_val[(int64_t)b * K + rank] = mv;
        }
    }
}

// The REFERENCE's order among equal scores, on the device (tie_order 1). evaluate.h:26-33 ranks a user's masked score row with
//     std::partial_sort_copy(index.begin(), index.end(), out, out + K, [&](int x1, int x2) { return ratings[x1] > ratings[x2]; })
// whose result among equal scores is the heap order of the C++ library's algorithm, not an order of the ids. The algorithm
// (libstdc++ bits/stl_algo.h __partial_sort_copy, bits/stl_heap.h __make_heap / __adjust_heap / __push_heap / __sort_heap --
// restated below operation for operation, every comparison the reference's comparator on the two scores):
//   1. the first K ids go into the result; make_heap (the heap's top is the WORST kept score);
//   2. every later id x in ascending order: if score[x] > score[top], __adjust_heap(result, 0, K, x);
//   3. sort_heap.
// Only step 2 touches the catalogue, and an id changes the heap only if its score beats the top, which never decreases: ONE WAVE
// per user row tests 64 items (or the maxima of 64 sixteen-item tiles) per step with a ballot against the current top and runs
// the heap operations of the few set lanes in lane (= id) order -- the sequence of __adjust_heap calls, hence every position in
// the heap, is the serial algorithm's. The heap lives in LDS, executed redundantly by all 64 lanes (same addresses, same values:
// broadcast reads, no divergence). A catalogue scored chunk by chunk carries the heap from launch to launch in out_idx / out_val
// (heap order) and publishes the top's score as the next chunk's store threshold -- it IS the running K-th best score.
struct RefHeap {
    float *v; int *id;
    __device__ __forceinline__ void adjust(int hole, int len, float val, int vid) {      // std::__adjust_heap
        const int top_index = hole;
        int second = hole;
        while (second < (len - 1) / 2) {
            second = 2 * (second + 1);
            if (v[second] > v[second - 1]) second--;                                     // comp(first + second, first + (second - 1))
            v[hole] = v[second]; id[hole] = id[second];
            hole = second;
        }
        if ((len & 1) == 0 && second == (len - 2) / 2) {
            second = 2 * (second + 1);
            v[hole] = v[second - 1]; id[hole] = id[second - 1];
            hole = second - 1;
        }
        int parent = (hole - 1) / 2;                                                     // std::__push_heap
        while (hole > top_index && v[parent] > val) {                                    // comp(first + parent, value)
            v[hole] = v[parent]; id[hole] = id[parent];
            hole = parent;
            parent = (hole - 1) / 2;
        }
        v[hole] = val; id[hole] = vid;
    }
    __device__ __forceinline__ void make(int len) {                                      // std::__make_heap
        if (len < 2) return;
        int parent = (len - 2) / 2;
        while (true) {
            const float val = v[parent];
            const int vid = id[parent];
            adjust(parent, len, val, vid);
            if (parent == 0) return;
            parent--;
        }
    }
    __device__ __forceinline__ void sort(int len) {                                      // std::__sort_heap (__pop_heap per step)
        while (len > 1) {
            --len;
            const float val = v[len];
            const int vid = id[len];
            v[len] = v[0]; id[len] = id[0];
            adjust(0, len, val, vid);
        }
    }
};
__device__ __forceinline__ void wave_lds_sync() {         // LDS written by other lanes of this wave is read next
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float lane_value(float v, int lane) {      // v of a wave-uniform lane
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), __builtin_amdgcn_readfirstlane(lane)));
}

constexpr int REF_KMAX = 1024;
// scores: row b's items [item0, item0 + n) at scores[b * lds + (item - item0)]; mask_bits (nullable): bit (item) of row b set =
// masked (score -inf, cpp/uni_evaluator.py:149-154); TILES: tile_max[b * tmax_ld + t] = the maximum BEFORE masking of the range's
// t-th 16-item tile -- only tiles whose maximum beats the heap's top are read (the chunked scorer stores exactly the tiles whose
// maximum reaches the threshold this kernel published after the previous chunk, which the top never falls below).
// first: the range starts the catalogue (fill + make_heap; n >= K); last: sort_heap, out_* = the final lists; otherwise out_* keep
// the heap and thr[b] (nullable) = its top's score.
template <bool TILES>
__global__ __launch_bounds__(256) void ref_order_kernel(const float *__restrict__ scores, int64_t lds, int64_t n,
                                                        const float *__restrict__ tile_max, int64_t tmax_ld, int B, int K,
                                                        const uint32_t *__restrict__ mask_bits, int64_t bits_ld, int64_t item0,
                                                        int32_t *out_idx, float *out_val, int64_t ldo, float *thr, int first, int last) {
    extern __shared__ float ref_lds[];                   // [4 waves][K] scores, [4 waves][K] ids
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;                                  // (no workgroup barrier below)
    RefHeap h;
    h.v = ref_lds + wave * K;
    h.id = reinterpret_cast<int *>(ref_lds + 4 * K) + wave * K;
    const float *row = scores + (int64_t)b * lds;
    const uint32_t *bits = mask_bits ? mask_bits + (int64_t)b * bits_ld : nullptr;
    int32_t *oi = out_idx + (int64_t)b * ldo;
    float *ov = out_val ? out_val + (int64_t)b * ldo : nullptr;
    auto score_at = [&](int64_t i) -> float {            // i relative to item0
        const int64_t g = item0 + i;
        if (bits && ((bits[g >> 5] >> (g & 31)) & 1u)) return -INFINITY;
        return row[i];
    };
    int64_t start = 0;
    if (first) {
        for (int j = lane; j < K; j += 64) { h.v[j] = score_at(j); h.id[j] = (int)(item0 + j); }
        wave_lds_sync();
        h.make(K);
        start = K;
    } else {
        for (int j = lane; j < K; j += 64) { h.v[j] = ov[j]; h.id[j] = oi[j]; }
        wave_lds_sync();
    }
    float top = h.v[0];
    if (TILES) {
        const float *tmx = tile_max + (int64_t)b * tmax_ld;
        const int64_t n_tiles = (n + TI - 1) / TI;
        // (the maxima of 8 x 64 tiles are fetched at once: the loads of a chunk's 1024 tiles are two round trips, not sixteen
        // dependent ones -- the kernel is a chain of latencies, one wave per user)
        constexpr int PF = 8;
        for (int64_t tb0 = (start / TI) & ~(int64_t)63; tb0 < n_tiles; tb0 += 64 * PF) {
            float tmv[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int64_t t = tb0 + 64 * u + lane;
                tmv[u] = (t < n_tiles && t >= start / TI) ? tmx[t] : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
            const int64_t tb = tb0 + 64 * u;
            const float tm = tmv[u];
            unsigned long long m = __builtin_amdgcn_ballot_w64(tm > top);
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                if (!(lane_value(tm, j) > top)) continue;                 // the top has risen past this tile since the ballot
                const int64_t i = (tb + j) * TI + (lane & 15);
                const float v = (i < n && i >= start) ? score_at(i) : -INFINITY;
                unsigned long long m2 = __builtin_amdgcn_ballot_w64(v > top) & 0xFFFFull;
                while (m2) {
                    const int jj = __builtin_ctzll(m2);
                    m2 &= m2 - 1;
                    const float vj = lane_value(v, jj);
                    if (vj > top) {                                       // comp(first, result_first) at the serial algorithm's turn
                        h.adjust(0, K, vj, (int)(item0 + (tb + j) * TI + jj));
                        top = h.v[0];
                    }
                }
            }
            }
        }
    } else {
        constexpr int PF = 8;                              // 8 x 64 scores in flight per round trip
        for (int64_t base0 = start; base0 < n; base0 += 64 * PF) {
            float vv[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int64_t i = base0 + 64 * u + lane;
                vv[u] = i < n ? score_at(i) : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int64_t base = base0 + 64 * u;
                const float v = vv[u];
                unsigned long long m = __builtin_amdgcn_ballot_w64(v > top);
                while (m) {
                    const int j = __builtin_ctzll(m);
                    m &= m - 1;
                    const float vj = lane_value(v, j);
                    if (vj > top) {
                        h.adjust(0, K, vj, (int)(item0 + base + j));
                        top = h.v[0];
                    }
                }
            }
        }
    }
    if (last) h.sort(K);
    else if (thr && lane == 0) thr[b] = top;
    wave_lds_sync();
    for (int j = lane; j < K; j += 64) { oi[j] = h.id[j]; if (ov) ov[j] = h.v[j]; }
}

struct MetricIds { int id[8]; };

// metric.h:17-106. One wave per user, a lane per rank: the hit flags of the K ranks are found in parallel (ballots), the
// discount table 1/log2(i+2) once per workgroup, and every lane then evaluates its own prefix of the reference's
// sequential recurrences IN THE REFERENCE'S ORDER (float / double exactly where the C++ rounds), so the curves have the
// bits of a one-thread-per-metric loop at a fraction of its latency. K <= 256.
__global__ __launch_bounds__(256) void rank_metrics_kernel(const int32_t *__restrict__ rank, int B, int K, const int64_t *__restrict__ tptr,
                                                           const int32_t *__restrict__ titems, MetricIds mids, int n_metrics,
                                                           float *__restrict__ out) {
    __shared__ double s_w[RM_KMAX];
    __shared__ unsigned long long s_hit[4][RM_KMAX / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wv;
    for (int i = threadIdx.x; i < K; i += 256) s_w[i] = 1.0 / log2((double)(i + 2));
    const int KW = (K + 63) >> 6;
    int nt = 0;
    if (b < B) {
        const int32_t *r = rank + (int64_t)b * K;
        const int32_t *truth = titems + tptr[b];
        nt = (int)(tptr[b + 1] - tptr[b]);
        for (int c = 0; c < KW; ++c) {
            const int i = lane + 64 * c;
            const int x = i < K ? r[i] : -1;
            bool h = false;
            for (int j = 0; j < nt; ++j) h = h || (truth[j] == x);
            const unsigned long long bal = __ballot(h && i < K);
            if (lane == 0) s_hit[wv][c] = bal;
        }
    }
    __syncthreads();
    if (b >= B) return;
    // visit the hit ranks j <= i in ascending order
    auto for_hits_upto = [&](int i, auto &&fn) {
        for (int c = 0; c <= (i >> 6); ++c) {
            unsigned long long bits = s_hit[wv][c];
            if (c == (i >> 6)) bits &= (~0ull) >> (63 - (i & 63));
            while (bits) {
                const int j = __builtin_ctzll(bits) + 64 * c;
                bits &= bits - 1;
                fn(j);
            }
        }
    };
    for (int m = 0; m < n_metrics; ++m) {
        const int id = mids.id[m];
        float *o = out + ((int64_t)b * n_metrics + m) * K;
        for (int c = 0; c < KW; ++c) {
            const int i = lane + 64 * c;
            if (i >= K) continue;
            float v = 0.f;
            if (id == 1 || id == 2) {
                int hits = 0;
                for_hits_upto(i, [&](int) { ++hits; });
                v = id == 1 ? (float)(1.0 * hits / (i + 1)) : (float)(1.0 * hits / (double)nt);
            } else if (id == 3) {
                int hits = 0;
                float sum_pre = 0.f;
                for_hits_upto(i, [&](int j) { ++hits; sum_pre += (float)(1.0 * hits / (j + 1)); });
                v = hits == 0 ? 0.f : sum_pre / hits;
            } else if (id == 4) {
                float dcg = 0.f, idcg = 0.f;
                for_hits_upto(i, [&](int j) { dcg = (float)((double)dcg + s_w[j]); });
                const int lim = i < nt ? i + 1 : nt;
                for (int j = 0; j < lim; ++j) idcg = (float)((double)idcg + s_w[j]);
                v = dcg / idcg;
            } else if (id == 5) {
                int first = -1;
                for (int c2 = 0; c2 <= (i >> 6) && first < 0; ++c2) {
                    unsigned long long bits = s_hit[wv][c2];
                    if (c2 == (i >> 6)) bits &= (~0ull) >> (63 - (i & 63));
                    if (bits) first = __builtin_ctzll(bits) + 64 * c2;
                }
                v = first < 0 ? 0.f : (float)(1.0 / (first + 1));
            }
            o[i] = v;
        }
    }
}

}  // namespace elimrec

using namespace elimrec;

static int g_score_math = -1;
static int score_math() {
    if (g_score_math < 0) {
        const char *e = getenv("ELIMREC_EVAL_MATH");
        g_score_math = (e && (e[0] == 'e' || e[0] == '0')) ? 0 : 1;      // "exact" / "0": IEEE division + libm expf
    }
    return g_score_math;
}
static int g_score_b3 = 1;
static int score_bf16x3() { return g_score_b3; }
extern "C" void elimrec_score_set_bf16x3(int on) { g_score_b3 = on ? 1 : 0; }
extern "C" int elimrec_score_get_bf16x3(void) { return score_bf16x3(); }
extern "C" void elimrec_score_set_math(int mode) { g_score_math = mode ? 1 : 0; }
extern "C" int elimrec_score_get_math(void) { return score_math(); }

// ONE device word per process counting the waves that saw a score outside their launch's range invariant (ScoreArgs::lo / hi)
static int *g_range_flag = nullptr;
static int *score_range_flag() {
    if (!g_range_flag) {
        if (hipMalloc((void **)&g_range_flag, 256) != hipSuccess) { g_range_flag = nullptr; return nullptr; }
        (void)hipMemset(g_range_flag, 0, 256);
    }
    return g_range_flag;
}
// *h_count = waves that saw an out-of-range score since the last reset, read BEHIND everything enqueued on `stream` (it
// synchronises with it); reset != 0: the word is cleared behind the read.
extern "C" int elimrec_score_range_violations(int64_t *h_count, int reset, void *stream) {
    ELIMREC_REQUIRE(h_count, "score_range_violations: null pointer");
    *h_count = 0;
    if (!g_range_flag) return 0;                     // no scorer launch yet
    int v = 0;
    hipStream_t s = (hipStream_t)stream;
    int rc = check_hip(hipMemcpyAsync(&v, g_range_flag, sizeof(int), hipMemcpyDeviceToHost, s), "score_range_violations: copy");
    if (rc) return rc;
    if (reset) { rc = check_hip(hipMemsetAsync(g_range_flag, 0, sizeof(int), s), "score_range_violations: reset"); if (rc) return rc; }
    rc = check_hip(hipStreamSynchronize(s), "score_range_violations: synchronize");
    *h_count = v;
    return rc;
}

static void score_range_bounds(int predict_type, int fusion_mode, float *lo, float *hi) {      // sigma(1) = 0.7310586, sigma(-1) = 0.2689414
    const float slack = 1e-6f;
    *lo = 0.f; *hi = 1.f;
    if (predict_type == 0 || (fusion_mode == 0 && predict_type == 1)) { *lo = 0.5f - slack; *hi = 0.7310586f + slack; }
    else if (fusion_mode == 0 && predict_type == 2) { *lo = 0.2689414f - slack; *hi = 0.7310586f + slack; }
}

static inline int n_item_tiles(int64_t I) { return (int)((I + TI - 1) / TI); }   // 16-item tiles (the finest of the forms)

constexpr int64_t SCORE_PILOT = 2048;     // ... the first chunk of a chunked pass: its top-K gives every later launch a store threshold
constexpr int64_t SCORE_CHUNK = 16384;    // items per scorer launch when only top-K is wanted (a multiple of 512)

// Workspace layout of elimrec_score_topk. full: a [B x I] score block (the caller wants the scores, or a scorer without tile
// maxima runs); top-K only: a [B x SCORE_CHUNK] block, the tile maxima of one chunk and the per-chunk candidate lists.
struct ScoreLayout { size_t partial, mean, scores, flags, sqn, bits, tmax, cand_val, cand_idx, planes, total; };
// d > 0 with the chunked layout: room for one chunk's item rows as three bf16 planes (score_t16b_kernel, recdim 32 / 64)
static ScoreLayout score_layout(int B, int64_t U, int64_t I, int S, int K, bool topk_only, int d = 0) {
    const size_t b = (size_t)(B > 0 ? B : 1);
    const int64_t cols = topk_only && I > SCORE_CHUNK ? SCORE_CHUNK : I;
    ScoreLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += align_up(bytes, 256); return at; };
    L.partial = take((size_t)n_item_tiles(I) * b * sizeof(float));
    L.mean = take(b * sizeof(float));
    L.scores = take(b * (size_t)cols * sizeof(float));
    L.flags = take(b * sizeof(int32_t));
    L.sqn = take((size_t)(U + I) * (size_t)(1 + S) * sizeof(float));
    L.bits = take(b * (size_t)((I + 31) / 32) * sizeof(uint32_t));
    L.tmax = take(b * (size_t)n_item_tiles(cols) * sizeof(float));
    L.cand_val = take(topk_only ? b * (size_t)(K > 0 ? K : 1) * sizeof(float) : 0);      // the running list's scores when the caller keeps none
    L.cand_idx = take(topk_only ? b * sizeof(float) : 0);                                // the running K-th best score per user
    L.planes = off;
    const int64_t prow = I > SCORE_CHUNK ? SCORE_CHUNK : I;            // one chunk's item rows
    if (d == 32 || d == 64) take((size_t)prow * 3 * (size_t)(1 + S) * (size_t)d * 2 + (size_t)prow * (size_t)(S > 0 ? S : 1) * 4);
    L.total = off;
    return L;
}

extern "C" size_t elimrec_score_workspace(int B, int64_t I, int K) {      // (without the squared-norm table: legacy entry)
    return score_layout(B, 0, I, 0, K, false).total;
}

extern "C" size_t elimrec_score_workspace2(int B, int64_t U, int64_t I, int S, int K) {
    return score_layout(B, U, I, S, K, false).total;
}

extern "C" size_t elimrec_score_workspace_topk(int B, int64_t U, int64_t I, int S, int K) {
    return score_layout(B, U, I, S, K, true).total;
}

// The ONE predicate for the chunked (top-K only, no [B x I] score block) form, shared by the sizing function below and by
// elimrec_score_topk: the 16-user-per-wave scorer (recdim 32 / 64 / 128, 1..3 heads, MFMA + T16 forms enabled), K <= 256
// (the tile-guided selection with its running list) and more than one chunk -- any catalogue size.
constexpr size_t TOPK_MERGE_LDS_MAX = 160 * 1024;
// (the plain-VALU scorer, the one-user-tile-per-workgroup MFMA scorer without tile maxima and the unchunked top-K were run-time
// switches until round 6 -- ELIMREC_SCORE_VALU / _T16 / _CHUNKED / _RESIDENT; each lost to the form below at every shape measured,
// docs/REJECTED.md -- and are now fixed: the forms are chosen by recdim and call shape alone)
static bool score_uses_mfma() { return true; }
static bool score_uses_t16() { return true; }
static bool score_uses_chunks() { return true; }
static bool score_t16_path(int d, int S) {
    return score_uses_mfma() && score_uses_t16() && (d == 32 || d == 64 || d == 128) && S >= 1 && S <= 3;
}
static bool score_chunked_form(int d, int S, int K, int64_t I, bool want_scores, bool want_topk) {
    return score_t16_path(d, S) && score_uses_chunks() && !want_scores && want_topk && K <= RM_KMAX && I > SCORE_CHUNK;
}
static bool score_matrix_chunks(int d, int S, int64_t I) {
    return score_t16_path(d, S) && score_uses_chunks() && I > SCORE_CHUNK;
}

// Bytes elimrec_score_topk needs for THIS call shape: recdim d, K, and whether the caller passes a score matrix
// (want_scores) -- the chunked layout exactly when the call will take the chunked form, the full [B x I] layout otherwise
// (any recdim, any K: the reference accepts both, models/EliMRec.py:96-113, evaluator/backend/cpp/uni_evaluator.py:131).
extern "C" size_t elimrec_score_workspace_for(int B, int64_t U, int64_t I, int S, int K, int d, int want_scores) {
    return score_layout(B, U, I, S, K, score_chunked_form(d, S, K, I, want_scores != 0, K > 0), d).total;
}

extern "C" int elimrec_row_sqnorms(const float *d_Y, int64_t ldy, int64_t n_rows, int d, int n_blocks, float *d_out,
                                   void *stream) {
    ELIMREC_REQUIRE(d_Y && d_out && d > 0 && d % 4 == 0 && n_blocks >= 1 && ldy % 4 == 0, "row_sqnorms: bad arguments");
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_Y, ldy,
                       n_rows, d, n_blocks, d_out);
    ELIMREC_LAUNCH_CHECK("row_sqnorm");
    return 0;
}


// phase 0: the whole call. Item-sharded evaluation (this rank holds items [id_offset, id_offset + I) of I_total):
// phase 1 = pass 1 only, d_row_sum[b] = sum over MY items of sigmoid(u.i) (TIE; a no-op otherwise); the caller adds the
// shards' sums (all_reduce); phase 2 = the rest with row mean = d_row_sum[b] / I_total, top-K ids + id_offset.
static int score_topk_impl(const float *d_Y, int64_t ldy, int64_t U, int64_t I, const int64_t *d_users, int B,
                           int d, int S, uint32_t head_mask, int fusion_mode, int predict_type,
                           const float *d_sqnorm, const int64_t *d_train_ptr, const int32_t *d_train_items,
                           float *d_scores, int64_t lds, int K, int32_t *d_topk_idx, float *d_topk_val,
                           void *d_workspace, size_t workspace_bytes, void *stream, int phase, float *d_row_sum,
                           int64_t I_total, int64_t id_offset, int tie_order) {
    ELIMREC_REQUIRE(d_Y && d_users && d_workspace, "score_topk: null pointer");
    ELIMREC_REQUIRE(tie_order == 0 || tie_order == 1, "score_topk: tie_order 0 (score desc, id asc) or 1 (the reference's partial_sort_copy)");
    ELIMREC_REQUIRE(tie_order == 0 || (phase == 0 && K <= REF_KMAX),
                    "score_topk: the reference's tie order needs the whole catalogue in one call (no item shard) and K <= %d", REF_KMAX);
    ELIMREC_REQUIRE(phase == 0 || (d_row_sum && I_total >= I), "score_topk: sharded phases need d_row_sum and I_total >= I");
    if (phase == 1 && predict_type != 2) return 0;             // only TIE has a catalogue-wide mean
    ELIMREC_REQUIRE(d > 0 && d % 4 == 0 && ldy % 4 == 0, "score_topk: recdim/ldy must be multiples of 4");
    ELIMREC_REQUIRE(S >= 0 && S <= kMaxS, "score_topk: at most %d single-modal heads", kMaxS);
    ELIMREC_REQUIRE(fusion_mode >= 0 && fusion_mode <= 2 && predict_type >= 0 && predict_type <= 2,
                    "score_topk: bad fusion_mode/predict_type");
    ELIMREC_REQUIRE(d_scores || d_topk_idx || phase == 1, "score_topk: nothing to output");
    ELIMREC_REQUIRE(!d_topk_idx || (K > 0 && K <= I), "score_topk: need 0 < K <= I");
    if (B <= 0) return 0;
    const bool t16_path = score_t16_path(d, S);
    // only top-K wanted: no [B x I] score block -- the catalogue goes through the scorer in chunks (a workspace sized by
    // elimrec_score_workspace_for is enough; a larger one is accepted)
    const bool chunked = score_chunked_form(d, S, K, I, d_scores != nullptr, d_topk_idx != nullptr);
    // the score MATRIX of a catalogue beyond one chunk goes through the same launches, chunk by chunk, every tile stored: the rows
    // predict() returns are then the bits the chunked top-K ranks (one score form per call shape, whatever the caller asks for)
    const bool matrix_chunks = !chunked && score_matrix_chunks(d, S, I);      // (a [B x I] block, the caller's or -- K > 256 -- a private one)
    ScoreLayout L = score_layout(B, U, I, S, K, chunked);
    if (phase == 1) {        // row sums only: no score block is touched -- the chunked layout (no [B x I] block) will do as well
        const ScoreLayout Lc = score_layout(B, U, I, S, K, true);
        if (Lc.total < L.total) L = Lc;
    }
    if (workspace_bytes < L.total) {
        set_error("score_topk: workspace too small (%zu < %zu)", workspace_bytes, L.total);
        return ELIMREC_E_WORKSPACE;
    }
    // FAST math, chunked top-K, recdim 32 / 64: pass 2 on the bf16 matrix cores from exact three-piece splits (needs room for one
    // chunk's pieces behind the chunked layout: a workspace sized by elimrec_score_workspace_for has it)
    const int use_b3 = score_bf16x3();
    const ScoreLayout Lp = score_layout(B, U, I, S, K, chunked, d);
    const bool bf16x3 = (chunked || matrix_chunks) && use_b3 && score_math() == 1 && (d == 32 || d == 64) && phase != 1 && workspace_bytes >= Lp.total;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = n_item_tiles(I);
    char *ws = (char *)d_workspace;
    float *partial = (float *)(ws + L.partial);
    float *mean = (float *)(ws + L.mean);
    float *wscores = (float *)(ws + L.scores);
    int32_t *fallback = (int32_t *)(ws + L.flags);
    float *cand_val = (float *)(ws + L.cand_val);
    int32_t *cand_idx = (int32_t *)(ws + L.cand_idx);
    ScoreArgs a;
    a.Y = d_Y; a.ldy = ldy; a.U = U; a.I = I; a.users = d_users; a.B = B; a.d = d; a.S = S; a.head_mask = head_mask;
    a.fusion_mode = fusion_mode; a.predict_type = predict_type; a.row_mean = mean; a.partial = partial;
    a.scores = d_scores ? d_scores : wscores; a.lds = d_scores ? lds : (chunked ? SCORE_CHUNK : I);
    a.item0 = 0; a.item_end = I; a.planes = nullptr; a.inrm = nullptr; a.thr = nullptr;
    score_range_bounds(predict_type, fusion_mode, &a.lo, &a.hi);      // the range invariant of the (predict type, fusion mode)
    a.range_flag = score_range_flag();
    float *wsqn = (float *)(ws + L.sqn);
    if (!d_sqnorm && predict_type != 0) {            // not supplied: compute the whole table for this call
        const int64_t N = U + I;
        hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, d_Y, ldy, N, d, 1 + S, wsqn);
        ELIMREC_LAUNCH_CHECK("row_sqnorm");
    }
    a.sqn = d_sqnorm ? d_sqnorm : wsqn;
    const bool own_pass1 = predict_type == 2 && phase != 2;      // phase 2: the mean comes from the all-reduced sums
    float *mean_dst = phase == 1 ? d_row_sum : mean;            // phase 1: sums (divisor 1), straight to the caller
    const int64_t mean_div = phase == 1 ? 1 : I;
    if (phase == 2 && predict_type == 2) {
        hipLaunchKernelGGL(mean_from_sum_kernel, dim3((B + 255) / 256), dim3(256), 0, s, (const float *)d_row_sum, B, I_total, mean);
        ELIMREC_LAUNCH_CHECK("mean_from_sum");
    }
    uint32_t *wbits = (uint32_t *)(ws + L.bits);
    float *wtmax = (float *)(ws + L.tmax);
    const int64_t bits_ld = (I + 31) / 32;
    a.tile_max = nullptr; a.tmax_ld = n_item_tiles(chunked ? SCORE_CHUNK : I);
    bool tiles_ready = false;
    ELIMREC_REQUIRE(chunked || a.lds >= I, "score_topk: lds < I");
    if (t16_path) {
        // 16 users per wave, 128 per workgroup, a persistent grid over 16-item tiles (two workgroups per CU)
        const int t16 = (int)((I + TI - 1) / TI);
        const bool fast = score_math() == 1;
        if (d_topk_idx && K <= 256) {    // the selection reads the scorer's tile maxima and a bitmap of the masked items
            a.tile_max = wtmax;
            tiles_ready = true;
            if (d_train_ptr) {
                ELIMREC_REQUIRE(d_train_items, "score_topk: train_items missing");
                hipLaunchKernelGGL(train_bits_kernel, dim3(B), dim3(256), 0, s, d_train_ptr, d_train_items, I, wbits, bits_ld);
                ELIMREC_LAUNCH_CHECK("train_bits");
            }
        }
        auto t16_grid = [&](int pass) {      // chunks dealt evenly to at most 512 workgroups
            const int chunks = (t16 + t16_sub(pass) - 1) / t16_sub(pass);
            const int per = (chunks + 511) / 512;
            return dim3((unsigned)((chunks + per - 1) / per), (B + TW * TU - 1) / (TW * TU));
        };
        const dim3 grid1 = t16_grid(1), grid = t16_grid(2);
        auto t16_lds = [d](int pass, int nb) {
            const int cols = (pass == 1 ? 1 : nb) * d;
            return ((size_t)2 * t16_sub(pass) * TI * (cols + 4) + (size_t)TW * TU * (nb > 1 ? nb - 1 : 1) + TW * TU) * sizeof(float);
        };
#define ELIMREC_T16_LAUNCH_D(PASS, NB, PT, FM, FAST, GRID, DD)                                               \
    do {                                                                                                   \
        static bool attr = false;                                                                          \
        if (!attr && t16_lds(PASS, NB) > 64 * 1024) {                                                      \
            (void)hipFuncSetAttribute((const void *)score_t16_kernel<PASS, NB, PT, FM, FAST, DD>,          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)t16_lds(PASS, NB)); \
            attr = true;                                                                                   \
        }                                                                                                  \
        hipLaunchKernelGGL((score_t16_kernel<PASS, NB, PT, FM, FAST, DD>), GRID, dim3(512), t16_lds(PASS, NB), s, a, t16); \
    } while (0)
#define ELIMREC_T16_LAUNCH(PASS, NB, PT, FM, FAST, GRID)                                                     \
    do {                                                                                                   \
        if (d == 64) ELIMREC_T16_LAUNCH_D(PASS, NB, PT, FM, FAST, GRID, 64);                               \
        else if (d == 128) ELIMREC_T16_LAUNCH_D(PASS, NB, PT, FM, FAST, GRID, 128);                        \
        else ELIMREC_T16_LAUNCH_D(PASS, NB, PT, FM, FAST, GRID, 32);                                       \
    } while (0)
#define ELIMREC_T16_P2(NB, PT, FM)                                                                           \
    do {                                                                                                   \
        if (fast) ELIMREC_T16_LAUNCH(2, NB, PT, FM, true, grid);                                           \
        else ELIMREC_T16_LAUNCH(2, NB, PT, FM, false, grid);                                               \
    } while (0)
#define ELIMREC_T16_PASS1(NB)                                                                               \
    do {                                                                                                   \
        if (fast) ELIMREC_T16_LAUNCH(1, NB, -1, -1, true, grid1);                                          \
        else ELIMREC_T16_LAUNCH(1, NB, -1, -1, false, grid1);                                              \
    } while (0)
#define ELIMREC_T16_PASS2(NB)                                                                               \
    do {                                                                                                   \
        if (predict_type == 0) ELIMREC_T16_P2(NB, 0, 0);                                                   \
        else if (predict_type == 1 && fusion_mode == 0) ELIMREC_T16_P2(NB, 1, 0);                          \
        else if (predict_type == 1 && fusion_mode == 1) ELIMREC_T16_P2(NB, 1, 1);                          \
        else if (predict_type == 1) ELIMREC_T16_P2(NB, 1, 2);                                              \
        else if (fusion_mode == 0) ELIMREC_T16_P2(NB, 2, 0);                                               \
        else if (fusion_mode == 1) ELIMREC_T16_P2(NB, 2, 1);                                               \
        else ELIMREC_T16_P2(NB, 2, 2);                                                                     \
    } while (0)
        // pass 1 (TIE): row means of sigmoid(u.i) over the WHOLE catalogue
        a.item0 = 0; a.item_end = I;
        if (own_pass1 && !(bf16x3 && phase == 0)) {
            if (S == 1) ELIMREC_T16_PASS1(2);
            else if (S == 2) ELIMREC_T16_PASS1(3);
            else ELIMREC_T16_PASS1(4);
            ELIMREC_LAUNCH_CHECK("score_t16_pass1");
            hipLaunchKernelGGL(row_mean_kernel, dim3(B), dim3(256), 0, s, partial, (int)grid1.x, B, mean_div, mean_dst);
            ELIMREC_LAUNCH_CHECK("row_mean");
        }
        if (phase == 1) return 0;
        // pass 2 over items [a.item0, a.item_end): the arguments, tile count and grid are the lambda's (they shadow the outer ones)
        auto pass2 = [&](const ScoreArgs &a, int t16, dim3 grid) -> int {
            if (S == 1) ELIMREC_T16_PASS2(2);
            else if (S == 2) ELIMREC_T16_PASS2(3);
            else ELIMREC_T16_PASS2(4);
            ELIMREC_LAUNCH_CHECK("score_t16_pass2");
            return 0;
        };
        auto b3_lds = [d](int pass, int nb) {
            const int nh = pass == 1 ? 1 : nb;
            return ((size_t)2 * t16b_win(pass, nb, d) * TI * (3 * nh * d / 8 + 1) * 16 + ((size_t)TWB * TU * (nb > 1 ? nb - 1 : 1) + TWB * TU) * sizeof(float));
        };
#define ELIMREC_T16B_LAUNCH1(NB, GRID, DD)                                                                   \
    hipLaunchKernelGGL((score_t16b_kernel<1, NB, -1, -1, DD>), GRID, dim3(NTB), b3_lds(1, NB), s, a, t16)
#define ELIMREC_T16B_LAUNCH(NB, PT, FM, GRID, DD)                                                            \
    do {                                                                                                   \
        static bool attr = false;                                                                          \
        if (!attr && b3_lds(2, NB) > 64 * 1024) {                                                          \
            (void)hipFuncSetAttribute((const void *)score_t16b_kernel<2, NB, PT, FM, DD>,                  \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)b3_lds(2, NB));     \
            attr = true;                                                                                   \
        }                                                                                                  \
        hipLaunchKernelGGL((score_t16b_kernel<2, NB, PT, FM, DD>), GRID, dim3(NTB), b3_lds(2, NB), s, a, t16); \
    } while (0)
#define ELIMREC_T16B_D(NB, PT, FM, GRID)                                                                     \
    do {                                                                                                   \
        if (d == 64) ELIMREC_T16B_LAUNCH(NB, PT, FM, GRID, 64);                                            \
        else ELIMREC_T16B_LAUNCH(NB, PT, FM, GRID, 32);                                                    \
    } while (0)
#define ELIMREC_T16B_PASS2(NB)                                                                              \
    do {                                                                                                   \
        if (predict_type == 0) ELIMREC_T16B_D(NB, 0, 0, grid);                                             \
        else if (predict_type == 1 && fusion_mode == 0) ELIMREC_T16B_D(NB, 1, 0, grid);                    \
        else if (predict_type == 1 && fusion_mode == 1) ELIMREC_T16B_D(NB, 1, 1, grid);                    \
        else if (predict_type == 1) ELIMREC_T16B_D(NB, 1, 2, grid);                                        \
        else if (fusion_mode == 0) ELIMREC_T16B_D(NB, 2, 0, grid);                                         \
        else if (fusion_mode == 1) ELIMREC_T16B_D(NB, 2, 1, grid);                                         \
        else ELIMREC_T16B_D(NB, 2, 2, grid);                                                               \
    } while (0)
        auto pass1_b3 = [&](const ScoreArgs &a, int t16, dim3 grid) -> int {
            if (d == 64) { if (S == 1) ELIMREC_T16B_LAUNCH1(2, grid, 64); else if (S == 2) ELIMREC_T16B_LAUNCH1(3, grid, 64); else ELIMREC_T16B_LAUNCH1(4, grid, 64); }
            else { if (S == 1) ELIMREC_T16B_LAUNCH1(2, grid, 32); else if (S == 2) ELIMREC_T16B_LAUNCH1(3, grid, 32); else ELIMREC_T16B_LAUNCH1(4, grid, 32); }
            ELIMREC_LAUNCH_CHECK("score_t16b_pass1");
            return 0;
        };
        auto pass2_b3 = [&](const ScoreArgs &a, int t16, dim3 grid) -> int {
            if (S == 1) ELIMREC_T16B_PASS2(2);
            else if (S == 2) ELIMREC_T16B_PASS2(3);
            else ELIMREC_T16B_PASS2(4);
            ELIMREC_LAUNCH_CHECK("score_t16b_pass2");
            return 0;
        };
        if (chunked || matrix_chunks) {
            // the catalogue goes through the scorer SCORE_CHUNK items at a time. chunked (only top-K wanted): a [B x SCORE_CHUNK]
            // block instead of [B x I], a RUNNING top-K per user; matrix_chunks (the caller's [B x I] matrix): the same launches
            // with every tile stored at its place in the matrix -- the same bits
            const int nch = (int)((I + SCORE_CHUNK - 1) / SCORE_CHUNK);
            if (own_pass1 && bf16x3 && phase == 0) {
                // pass 1 (row means of sigmoid(u.i)) on the bf16 matrix cores too: chunk by chunk (the fused block's pieces only),
                // <= 64 workgroups per user group and chunk, partials in (chunk, workgroup) order
                uint4 *planes = (uint4 *)(ws + Lp.planes);
                const int ug = (B + TWB * TU - 1) / (TWB * TU);
                int n_part = 0;
                for (int c = 0; c < nch; ++c) {
                    ScoreArgs ac = a;
                    ac.item0 = (int64_t)c * SCORE_CHUNK;
                    ac.item_end = ac.item0 + SCORE_CHUNK < I ? ac.item0 + SCORE_CHUNK : I;
                    const int64_t cnt = ac.item_end - ac.item0;
                    const int tc = (int)((cnt + TI - 1) / TI);
                    hipLaunchKernelGGL(split3_items_kernel, dim3((unsigned)((cnt * (d / 8) + 255) / 256)), dim3(256), 0, s, d_Y, ldy, U,
                                       ac.item0, cnt, d, planes, (const float *)nullptr, 0, (float *)nullptr);
                    ELIMREC_LAUNCH_CHECK("split3_items(pass 1)");
                    int gx = tc < 64 ? tc : 64;
                    ac.planes = planes;
                    ac.partial = partial + (int64_t)n_part * B;
                    int rc = pass1_b3(ac, tc, dim3((unsigned)gx, ug));
                    if (rc) return rc;
                    n_part += gx;
                }
                hipLaunchKernelGGL(row_mean_kernel, dim3(B), dim3(256), 0, s, partial, n_part, B, mean_div, mean_dst);
                ELIMREC_LAUNCH_CHECK("row_mean");
            }
            // pass 2 + selection, chunk by chunk, with a RUNNING top-K per user: every chunk's selection leaves the best K so far and
            // their K-th score; the next chunk's scorer stores a tile's scores of a user only where the tile's maximum reaches that
            // score -- about one tile in seven after the 2 048-item pilot chunk, one in a hundred after the first full chunk -- so
            // the [B x 16 384] block is hardly written at all (it was 537 MB per launch at 8 192 users). The lists after the last
            // chunk are the result: no merge launch. tie_order 0: topk_tiles_kernel, lists by (score desc, id asc); tie_order 1:
            // ref_order_kernel, the reference's heap carried from chunk to chunk (its top IS the K-th best score so far).
            float *thrbuf = (float *)cand_idx;                           // [B]
            float *run_val = d_topk_val ? d_topk_val : cand_val;         // [B x K] (the caller's list is the running list)
            int64_t at = 0;
            for (int c = 0; at < I; ++c) {
                ScoreArgs ac = a;
                ac.item0 = at;
                const int64_t len = (chunked && c == 0 && I > SCORE_PILOT) ? SCORE_PILOT : SCORE_CHUNK;
                ac.item_end = ac.item0 + len < I ? ac.item0 + len : I;
                at = ac.item_end;
                if (chunked) ac.thr = c == 0 ? nullptr : thrbuf;
                else {                                                   // this chunk's columns of the [B x I] block
                    ac.scores = a.scores + ac.item0;
                    ac.lds = a.lds;
                    ac.thr = nullptr;
                    if (a.tile_max) ac.tile_max = a.tile_max + ac.item0 / TI;
                }
                const int64_t cnt = ac.item_end - ac.item0;
                const int tc = (int)((cnt + TI - 1) / TI);
                // ~512 workgroups per launch in all: a workgroup walks several tiles with its users' operands resident
                const int wu = (bf16x3 ? TWB : TW) * TU;               // users per workgroup
                const int ug = (B + wu - 1) / wu;
                int gx = 512 / ug > 0 ? 512 / ug : 1;
                if (gx > tc) gx = tc;
                const int per = (tc + gx - 1) / gx;
                int rc = 0;
                if (bf16x3) {
                    const int cols = (1 + S) * d;
                    uint4 *planes = (uint4 *)(ws + Lp.planes);
                    // (the chunk's inverse item norms behind its planes; predict type normal has no heads to normalise)
                    float *inrm = (float *)(ws + Lp.planes + (size_t)(I > SCORE_CHUNK ? SCORE_CHUNK : I) * 3 * (size_t)cols * 2);
                    hipLaunchKernelGGL(split3_items_kernel, dim3((unsigned)((cnt * (cols / 8) + 255) / 256)), dim3(256), 0, s, d_Y, ldy, U,
                                       ac.item0, cnt, cols, planes, predict_type != 0 ? a.sqn : (const float *)nullptr, 1 + S,
                                       predict_type != 0 ? inrm : (float *)nullptr);
                    ELIMREC_LAUNCH_CHECK("split3_items");
                    ac.planes = planes;
                    ac.inrm = inrm;
                    rc = pass2_b3(ac, tc, dim3((unsigned)((tc + per - 1) / per), ug));
                } else {
                    rc = pass2(ac, tc, dim3((unsigned)((tc + per - 1) / per), ug));
                }
                if (rc) return rc;
                if (!chunked) continue;
                if (tie_order == 1) {
                    hipLaunchKernelGGL(ref_order_kernel<true>, dim3((unsigned)((B + 3) / 4)), dim3(256), (size_t)K * 32, s,
                                       (const float *)ac.scores, ac.lds, cnt, (const float *)wtmax, ac.tmax_ld, B, K,
                                       d_train_ptr ? (const uint32_t *)wbits : (const uint32_t *)nullptr, bits_ld, ac.item0, d_topk_idx,
                                       run_val, (int64_t)K, thrbuf, c == 0 ? 1 : 0, at >= I ? 1 : 0);
                    ELIMREC_LAUNCH_CHECK("ref_order(chunk)");
                } else {
                    hipLaunchKernelGGL(topk_tiles_kernel, dim3(B), dim3(256), 0, s, ac.scores, ac.lds, cnt, (const float *)wtmax, ac.tmax_ld,
                                       tc, K, d_train_ptr, d_train_ptr ? (const uint32_t *)wbits : (const uint32_t *)nullptr, bits_ld,
                                       d_topk_idx, run_val, fallback, ac.item0, (int64_t)K, (int64_t)0, thrbuf, c == 0 ? 1 : 0);
                    ELIMREC_LAUNCH_CHECK("topk_tiles(chunk)");
                }
            }
            if (chunked) {
                if (a.range_flag) {
                    hipLaunchKernelGGL(range_check_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, (const float *)run_val,
                                       (const int32_t *)d_topk_idx, B, K, a.lo, a.hi, predict_type == 2 ? (const float *)mean : (const float *)nullptr,
                                       a.range_flag);
                    ELIMREC_LAUNCH_CHECK("range_check");
                }
                if (id_offset) {
                    hipLaunchKernelGGL(add_id_offset_kernel, dim3((unsigned)(((int64_t)B * K + 255) / 256)), dim3(256), 0, s, d_topk_idx,
                                       (int64_t)B * K, (int32_t)id_offset);
                    ELIMREC_LAUNCH_CHECK("add_id_offset");
                }
                return 0;
            }
        } else { int rc = pass2(a, t16, grid); if (rc) return rc; }
#undef ELIMREC_T16
#undef ELIMREC_T16_P2
#undef ELIMREC_T16_LAUNCH
#undef ELIMREC_T16_LAUNCH_D
    } else {
        dim3 grid(tiles, (B + MU - 1) / MU);
        if (own_pass1) {
            hipLaunchKernelGGL(score_mfma_kernel<1>, grid, dim3(256), 0, s, a);
            ELIMREC_LAUNCH_CHECK("score_mfma_pass1");
            hipLaunchKernelGGL(row_mean_kernel, dim3(B), dim3(256), 0, s, partial, tiles, B, mean_div, mean_dst);
            ELIMREC_LAUNCH_CHECK("row_mean");
        }
        if (phase == 1) return 0;
        hipLaunchKernelGGL(score_mfma_kernel<2>, grid, dim3(256), 0, s, a);
        ELIMREC_LAUNCH_CHECK("score_mfma_pass2");
    }
    if (d_train_ptr && (d_scores || !tiles_ready)) {       // the caller's score matrix is masked; a private one only if a sweep reads it
        ELIMREC_REQUIRE(d_train_items, "score_topk: train_items missing");
        hipLaunchKernelGGL(mask_train_kernel, dim3(B), dim3(128), 0, s, a.scores, a.lds, d_train_ptr, d_train_items, B);
        ELIMREC_LAUNCH_CHECK("mask_train");
    }
    if (d_topk_idx && tie_order == 1) {
        // the reference's heap over the whole row: guided by the tile maxima where the scorer left them (the private matrix is
        // then unmasked: the bitmap), item by item otherwise (the matrix is masked)
        if (tiles_ready)
            hipLaunchKernelGGL(ref_order_kernel<true>, dim3((unsigned)((B + 3) / 4)), dim3(256), (size_t)K * 32, s, (const float *)a.scores,
                               a.lds, I, (const float *)wtmax, a.tmax_ld, B, K, d_train_ptr ? (const uint32_t *)wbits : (const uint32_t *)nullptr,
                               bits_ld, (int64_t)0, d_topk_idx, d_topk_val, (int64_t)K, (float *)nullptr, 1, 1);
        else
            hipLaunchKernelGGL(ref_order_kernel<false>, dim3((unsigned)((B + 3) / 4)), dim3(256), (size_t)K * 32, s, (const float *)a.scores,
                               a.lds, I, (const float *)nullptr, (int64_t)0, B, K, (const uint32_t *)nullptr, (int64_t)0, (int64_t)0,
                               d_topk_idx, d_topk_val, (int64_t)K, (float *)nullptr, 1, 1);
        ELIMREC_LAUNCH_CHECK("ref_order");
        if (id_offset) {
            hipLaunchKernelGGL(add_id_offset_kernel, dim3((unsigned)(((int64_t)B * K + 255) / 256)), dim3(256), 0, s, d_topk_idx,
                               (int64_t)B * K, (int32_t)id_offset);
            ELIMREC_LAUNCH_CHECK("add_id_offset");
        }
    } else if (d_topk_idx) {
        int G = 32;
        while (G < 2 * K && G < 1024) G *= 2;
        if (tiles_ready) {
            hipLaunchKernelGGL(topk_tiles_kernel, dim3(B), dim3(256), 0, s, a.scores, a.lds, I, (const float *)wtmax, a.tmax_ld,
                               (int)a.tmax_ld, K, d_train_ptr, d_train_ptr ? (const uint32_t *)wbits : (const uint32_t *)nullptr,
                               bits_ld, d_topk_idx, d_topk_val, fallback, (int64_t)0, (int64_t)K, (int64_t)0, (float *)nullptr, 0);
            ELIMREC_LAUNCH_CHECK("topk_tiles");
        } else if (2 * K <= 1024) {
            hipLaunchKernelGGL(topk_select_kernel, dim3(B), dim3(1024), 0, s, a.scores, a.lds, I, K, G, d_topk_idx,
                               d_topk_val, fallback);
            ELIMREC_LAUNCH_CHECK("topk_select");
            hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(1024), 0, s, a.scores, a.lds, I, K, d_topk_idx, d_topk_val,
                               (const int32_t *)fallback, (const uint32_t *)nullptr, (int64_t)0, (int64_t)0, (int64_t)K, (int64_t)0);
        } else {
            hipLaunchKernelGGL(topk_kernel, dim3(B), dim3(1024), 0, s, a.scores, a.lds, I, K, d_topk_idx, d_topk_val,
                               (const int32_t *)nullptr, (const uint32_t *)nullptr, (int64_t)0, (int64_t)0, (int64_t)K, (int64_t)0);
        }
        ELIMREC_LAUNCH_CHECK("topk");
        if (id_offset) {
            hipLaunchKernelGGL(add_id_offset_kernel, dim3((unsigned)(((int64_t)B * K + 255) / 256)), dim3(256), 0, s, d_topk_idx,
                               (int64_t)B * K, (int32_t)id_offset);
            ELIMREC_LAUNCH_CHECK("add_id_offset");
        }
    }
    if (a.range_flag && t16_path && ((d_topk_idx && d_topk_val) || predict_type == 2)) {
        const bool lists = d_topk_idx && d_topk_val;
        hipLaunchKernelGGL(range_check_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, lists ? (const float *)d_topk_val : (const float *)nullptr,
                           (const int32_t *)d_topk_idx, B, K, a.lo, a.hi, predict_type == 2 ? (const float *)mean : (const float *)nullptr, a.range_flag);
        ELIMREC_LAUNCH_CHECK("range_check");
    }
    return 0;
}

// The check every scoring call ends with, over lists the caller holds (d_topk_val / d_topk_idx [B x K]; d_row_mean [B] nullable):
// offending user rows are added to the counter elimrec_score_range_violations reads.
extern "C" int elimrec_score_range_check(const float *d_topk_val, const int32_t *d_topk_idx, int B, int K, int predict_type, int fusion_mode,
                                         const float *d_row_mean, void *stream) {
    ELIMREC_REQUIRE((d_topk_val && d_topk_idx && K > 0) || d_row_mean, "score_range_check: nothing to check");
    ELIMREC_REQUIRE(fusion_mode >= 0 && fusion_mode <= 2 && predict_type >= 0 && predict_type <= 2, "score_range_check: bad fusion_mode/predict_type");
    if (B <= 0) return 0;
    int *flag = score_range_flag();
    ELIMREC_REQUIRE(flag, "score_range_check: no device counter");
    float lo, hi;
    score_range_bounds(predict_type, fusion_mode, &lo, &hi);
    hipLaunchKernelGGL(range_check_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_topk_val, d_topk_idx, B, K, lo, hi,
                       d_row_mean, flag);
    ELIMREC_LAUNCH_CHECK("range_check");
    return 0;
}

extern "C" int elimrec_score_topk(const float *d_Y, int64_t ldy, int64_t U, int64_t I, const int64_t *d_users, int B,
                                  int d, int S, uint32_t head_mask, int fusion_mode, int predict_type,
                                  const float *d_sqnorm, const int64_t *d_train_ptr, const int32_t *d_train_items,
                                  float *d_scores, int64_t lds, int K, int32_t *d_topk_idx, float *d_topk_val,
                                  void *d_workspace, size_t workspace_bytes, void *stream) {
    return score_topk_impl(d_Y, ldy, U, I, d_users, B, d, S, head_mask, fusion_mode, predict_type, d_sqnorm, d_train_ptr,
                           d_train_items, d_scores, lds, K, d_topk_idx, d_topk_val, d_workspace, workspace_bytes, stream, 0,
                           nullptr, I, 0, 0);
}

// ... with the order among equal scores chosen by the caller: tie_order 0 = (score descending, item id ascending), 1 = the
// reference's (evaluate.h:26-33: std::partial_sort_copy's heap order), replayed on the device by ref_order_kernel
extern "C" int elimrec_score_topk_ordered(const float *d_Y, int64_t ldy, int64_t U, int64_t I, const int64_t *d_users, int B,
                                          int d, int S, uint32_t head_mask, int fusion_mode, int predict_type,
                                          const float *d_sqnorm, const int64_t *d_train_ptr, const int32_t *d_train_items,
                                          float *d_scores, int64_t lds, int K, int32_t *d_topk_idx, float *d_topk_val,
                                          void *d_workspace, size_t workspace_bytes, int tie_order, void *stream) {
    return score_topk_impl(d_Y, ldy, U, I, d_users, B, d, S, head_mask, fusion_mode, predict_type, d_sqnorm, d_train_ptr,
                           d_train_items, d_scores, lds, K, d_topk_idx, d_topk_val, d_workspace, workspace_bytes, stream, 0,
                           nullptr, I, 0, tie_order);
}

// The reference's ranking of rows of (masked) scores that are already on the device: d_scores [n_rows x ld] -> d_topk_idx /
// d_topk_val (nullable) [n_rows x K], bit for bit std::partial_sort_copy's lists (one wave per row, ref_order_kernel).
extern "C" int elimrec_topk_reference_order_device(const float *d_scores, int64_t n_rows, int64_t I, int64_t ld, int K,
                                                   int32_t *d_topk_idx, float *d_topk_val, void *stream) {
    ELIMREC_REQUIRE(d_scores && d_topk_idx && I > 0 && K > 0 && K <= I && K <= REF_KMAX && ld >= I && I < INT32_MAX && n_rows < INT32_MAX,
                    "topk_reference_order_device: bad arguments (0 < K <= min(I, %d), ld >= I)", REF_KMAX);
    if (n_rows <= 0) return 0;
    hipLaunchKernelGGL(ref_order_kernel<false>, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), (size_t)K * 32, (hipStream_t)stream, d_scores,
                       ld, I, (const float *)nullptr, (int64_t)0, (int)n_rows, K, (const uint32_t *)nullptr, (int64_t)0, (int64_t)0,
                       d_topk_idx, d_topk_val, (int64_t)K, (float *)nullptr, 1, 1);
    ELIMREC_LAUNCH_CHECK("ref_order");
    return 0;
}

extern "C" int elimrec_score_topk_shard(const float *d_Y, int64_t ldy, int64_t U, int64_t I, const int64_t *d_users, int B,
                                        int d, int S, uint32_t head_mask, int fusion_mode, int predict_type,
                                        const float *d_sqnorm, const int64_t *d_train_ptr, const int32_t *d_train_items,
                                        float *d_scores, int64_t lds, int K, int32_t *d_topk_idx, float *d_topk_val,
                                        void *d_workspace, size_t workspace_bytes, int phase, float *d_row_sum,
                                        int64_t I_total, int64_t id_offset, void *stream) {
    ELIMREC_REQUIRE(phase == 1 || phase == 2, "score_topk_shard: phase 1 (row sums) or 2 (scores / top-K)");
    ELIMREC_REQUIRE(id_offset >= 0 && id_offset + I <= I_total && I_total < (int64_t)INT32_MAX, "score_topk_shard: bad item range");
    return score_topk_impl(d_Y, ldy, U, I, d_users, B, d, S, head_mask, fusion_mode, predict_type, d_sqnorm, d_train_ptr,
                           d_train_items, d_scores, lds, K, d_topk_idx, d_topk_val, d_workspace, workspace_bytes, stream, phase,
                           d_row_sum, I_total, id_offset, 0);
}

// Merge per-shard (or per-chunk) candidate lists: d_cand_val / d_cand_idx [B x n_cand] (idx < 0 = no candidate) -> the K
// best per row by (score descending, item id ascending), the rule of every selection in this file.
extern "C" int elimrec_topk_merge(const float *d_cand_val, const int32_t *d_cand_idx, int B, int n_cand, int K,
                                  int32_t *d_topk_idx, float *d_topk_val, void *stream) {
    ELIMREC_REQUIRE(d_cand_val && d_cand_idx && d_topk_idx && K >= 1 && n_cand >= 1, "topk_merge: bad arguments");
    ELIMREC_REQUIRE((size_t)n_cand * 8 <= TOPK_MERGE_LDS_MAX, "topk_merge: %d candidates per row exceed the LDS budget", n_cand);
    if (B <= 0) return 0;
    if ((size_t)n_cand * 8 > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)topk_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TOPK_MERGE_LDS_MAX);
    hipLaunchKernelGGL(topk_merge_kernel, dim3(B), dim3(256), (size_t)n_cand * 8, (hipStream_t)stream, d_cand_val, d_cand_idx, n_cand, K,
                       d_topk_idx, d_topk_val);
    ELIMREC_LAUNCH_CHECK("topk_merge");
    return 0;
}

extern "C" int elimrec_rank_metrics(const int32_t *d_topk_idx, int B, int K, const int64_t *d_truth_ptr,
                                    const int32_t *d_truth_items, const int *metric_ids, int n_metrics, float *d_out,
                                    void *stream) {
    ELIMREC_REQUIRE(d_topk_idx && d_truth_ptr && d_truth_items && metric_ids && d_out, "rank_metrics: null pointer");
    ELIMREC_REQUIRE(n_metrics >= 1 && n_metrics <= 8, "rank_metrics: 1..8 metrics");
    MetricIds mids;
    for (int m = 0; m < 8; ++m) {
        mids.id[m] = m < n_metrics ? metric_ids[m] : 0;
        ELIMREC_REQUIRE(m >= n_metrics || (metric_ids[m] >= 1 && metric_ids[m] <= 5), "rank_metrics: unknown metric id %d",
                        metric_ids[m]);
    }
    ELIMREC_REQUIRE(K >= 1 && K <= RM_KMAX, "rank_metrics: 1 <= K <= %d", RM_KMAX);
    if (B <= 0) return 0;
    hipLaunchKernelGGL(rank_metrics_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_topk_idx,
                       B, K, d_truth_ptr, d_truth_items, mids, n_metrics, d_out);
    ELIMREC_LAUNCH_CHECK("rank_metrics");
    return 0;
}


// The reference's OWN order among equal scores, for the rows that have any: evaluate.h:26-33 ranks a user's scores with
// std::partial_sort_copy over the item ids under comp(x1, x2) = ratings[x1] > ratings[x2], whose result among ties is the heap order
// of the C++ library's algorithm -- not an order of the ids. The device ranks by (score descending, id ascending); a caller that
// wants the reference's list bit for bit (--tie_order=reference) hands the few rows whose K + 1 best scores contain a tie to this
// HOST function, which runs that very algorithm of the C++ library this package is built with on the row's masked scores.
// h_scores [n_rows x ld] (host), h_topk [n_rows x K] (host, out).
#include <algorithm>
#include <numeric>
#include <vector>
extern "C" int elimrec_topk_reference_order(const float *h_scores, int64_t n_rows, int64_t I, int64_t ld, int K, int32_t *h_topk) {
    ELIMREC_REQUIRE(h_scores && h_topk && I > 0 && K > 0 && K <= I && ld >= I && I < INT32_MAX, "topk_reference_order: bad arguments");
    std::vector<int> index((size_t)I);
    for (int64_t r = 0; r < n_rows; ++r) {
        const float *ratings = h_scores + r * ld;
        std::iota(index.begin(), index.end(), 0);
        int32_t *out = h_topk + r * K;
        std::partial_sort_copy(index.begin(), index.end(), out, out + K, [ratings](int x1, int x2) -> bool { return ratings[x1] > ratings[x2]; });
    }
    return 0;
}
