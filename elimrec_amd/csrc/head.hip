// Everything the training step evaluates AFTER the graph, at the batch's active rows, in ONE launch
// (models/EliMRec.py:233-236 folded, :262-270, :146-151): for a tile of 32 active rows
//     Out_m = S_m[rows] W_m^T + c[rows] b_m^T + narrow[rows]         (feature blocks, folded form)
//     Y_0   = Out W_side^T + b_side                                   (embedding_{user,item}_after_GCN)
//     Y_m   = Out_m Ws_m^T + bs_m                                     (s_dense_m)
// with the row tile, the Out tile and the K-half partial sums in LDS and the weights read in MFMA-fragment order from a
// packed copy (one coalesced 256-B load per v_mfma_f32_32x32x2_f32, 16 steps in flight), so the two batched GEMM launches
// of the unfused path (each a latency chain of global -> LDS -> MFMA rounds at ~6 % MFMA utilisation) and the Out
// round trip through HBM between them disappear. fp32-input MFMA: an exact fp32 fma chain, K split in two halves that
// are added once -- within fp32 round-off of the unfused path (tests/test_shard_gpu.py).
// Specialised for recdim = 64 (two 32-column MFMA tiles) and feature widths that fit LDS; other shapes keep the
// batched-GEMM path.
#include "common.h"
#include <hip/hip_fp16.h>
#include "rows_args.h"
#include <cstdlib>

namespace elimrec {

constexpr int HD = 64;            // recdim
constexpr int HMAXM = 3;

struct PackJob { const float *W; int64_t ld; int K; int64_t dst; int N; int64_t sn, sk; };   // 16-row form: element (n, k) = W[n*sn + k*sk], N columns
struct PackJobs { PackJob j[16]; int n; int first_block[17]; };

struct HeadFwdArgs {
    const int32_t *act, *seg_info;
    const float *out0; int64_t ld_out0;             // [R x 64] block 0 of Out (layer means), compact
    const float *narrow; int64_t ld_nar;            // [R x 64] shared part, compact
    const float *c;                                 // [N]
    const float *S[HMAXM]; int64_t ldS[HMAXM]; int D[HMAXM];
    const float *bias_m[HMAXM];
    int n_mod;
    const float *pk;
    int64_t off_Wm[HMAXM], off_Wf[2], off_Ws[HMAXM];
    const float *bias_f[2], *bias_s[HMAXM];
    float *OutAct; int64_t ld_out;
    float *YAct; int64_t ld_y;
    int a_off[HMAXM];                               // LDS offsets (floats) of the S_m row tiles
    int out_off, part_off;
    int stage;                                      // head_fwd16_kernel: 0 = whole head; 1 = the feature blocks only, WITHOUT the
                                                    // shared part (needs nothing of the graph: can run beside the forward hops);
                                                    // 2 = the rest (shared part added to what stage 1 left in OutAct, fusion, heads)
    // 16-bit constants read where they lie (elimrec_head_fwd_fused_src16): rows [S_1 | .. | S_n | c_hi c_lo] of fp16 (1) / bf16 (2)
    // elements, widened in registers -- no separate widening pass over the batch's rows in front of the head. s_out / c_out
    // (nullable): the widened rows in active-row order, for the weight-gradient launches of the backward half.
    int sdtype; const uint16_t *tab16; int64_t row_elems; int s_off[HMAXM]; int c_off;
    float *s_out; int64_t ld_sout; float *c_out;
    // out0 / narrow as the forward exchange delivers them (elimrec_head_fwd_fused_peers): peer q's piece of row r, columns
    // [q*dl, (q+1)*dl), at out0 + q * peer_stride + r * ld_out0 -- the received [W][R][out0 dl | narrow dl] buffer read in place
    int peer_dl; int64_t peer_stride;
};

__device__ __forceinline__ float widen16(uint32_t h, int sdtype) {        // one fp16 / bf16 element (low 16 bits of h) -> fp32, exact
    if (sdtype == 2) return __uint_as_float(h << 16);
    const unsigned short hs = (unsigned short)h;
    return __half2float(*reinterpret_cast<const __half *>(&hs));
}
// four consecutive elements of feature table m of `node`, as fp32
__device__ __forceinline__ float4 head_s4(const HeadFwdArgs &a, int m, int64_t node, int c4) {
    if (a.sdtype == 0) return *reinterpret_cast<const float4 *>(a.S[m] + node * a.ldS[m] + 4 * c4);
    const uint2 w = *reinterpret_cast<const uint2 *>(a.tab16 + node * a.row_elems + a.s_off[m] + 4 * c4);
    return make_float4(widen16(w.x & 0xFFFFu, a.sdtype), widen16(w.x >> 16, a.sdtype), widen16(w.y & 0xFFFFu, a.sdtype), widen16(w.y >> 16, a.sdtype));
}
__device__ __forceinline__ float head_c(const HeadFwdArgs &a, int64_t node) {      // c = hi + lo in 16-bit storage
    if (a.sdtype == 0) return a.c[node];
    const uint32_t w = *reinterpret_cast<const uint32_t *>(a.tab16 + node * a.row_elems + a.c_off);
    return widen16(w & 0xFFFFu, a.sdtype) + widen16(w >> 16, a.sdtype);
}

// ---------------------------------------------------------------------------------------------------------------------
// The head on 16-row tiles: v_mfma_f32_16x16x4_f32, wave w owns output columns [16w, 16w + 16) of EVERY block over the whole
// K -- no K-split, no partial-sum exchange, every wave takes part in both epilogues, three workgroups per CU. (A 32-row form --
// v_mfma_f32_32x32x2_f32, K split over wave pairs, one workgroup per CU -- was the first fused head and a run-time switch until
// round 6: a chain of exposed latencies at one wave per SIMD, launch + LDS skeleton 12 us, MFMAs 9, gathers + stores 10; removed.)
// Packed weights: element (n, k) of a [64 x K] matrix at ((n / 16) * (K / 4) + k / 4) * 64 + (k & 3) * 16 + n % 16,
// i.e. the B operand of MFMA step s of column tile ct is the 64 consecutive floats at ((ct * K/4) + s) * 64.
typedef float v4h __attribute__((ext_vector_type(4)));
constexpr int H16 = 16;

__global__ __launch_bounds__(256) void pack_head_weights16_kernel(PackJobs jobs, float *__restrict__ pk) {
    int q = 0;
    while (q + 1 < jobs.n && (int)blockIdx.x >= jobs.first_block[q + 1]) ++q;
    const PackJob &jb = jobs.j[q];
    const int64_t total = (int64_t)jb.N * jb.K;
    const int nb = jobs.first_block[q + 1] - jobs.first_block[q];
    for (int64_t e = (int64_t)((int)blockIdx.x - jobs.first_block[q]) * 256 + threadIdx.x; e < total; e += (int64_t)nb * 256) {
        const int lane = (int)(e & 63);
        const int64_t blk = e >> 6;                 // ct * (K/4) + s
        const int ct = (int)(blk / (jb.K / 4)), s = (int)(blk - (int64_t)ct * (jb.K / 4));
        pk[jb.dst + e] = jb.W[(int64_t)(ct * 16 + (lane & 15)) * jb.sn + (int64_t)(4 * s + (lane >> 4)) * jb.sk];
    }
}

// acc += A[16 x 4*nsteps] . B : A from LDS (ap = this lane's row and k-quarter), B = packed weights (bp = column tile + lane)
__device__ __forceinline__ v4h head16_run(v4h acc, const float *ap, const float *__restrict__ bp, int nsteps) {
    constexpr int PF = 16;
    nsteps = __builtin_amdgcn_readfirstlane(nsteps);
    float bq[PF];
    if (nsteps >= PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) bq[u] = bp[u * 64];
    }
    int s = 0;
    for (; s + PF <= nsteps; s += PF) {
        float bc[PF], av[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) bc[u] = bq[u];
        if (s + 2 * PF <= nsteps) {
#pragma unroll
            for (int u = 0; u < PF; ++u) bq[u] = bp[(int64_t)(s + PF + u) * 64];
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) av[u] = ap[4 * (s + u)];
#pragma unroll
        for (int u = 0; u < PF; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bc[u], acc, 0, 0, 0);
    }
    for (; s < nsteps; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s], bp[(int64_t)s * 64], acc, 0, 0, 0);
    return acc;
}

// ROWS: stage 2 with out0 / narrow EVALUATED here (rows_piece: the layer means of the tile's 16 rows, hop L inline) instead of
// read back from a rows launch -- thread (row tid / 16, float4 column tid % 16), the mapping of slab_rows_kernel<16>; recdim 64.
template <bool ROWS>
__device__ __forceinline__ void head_fwd16_body(const HeadFwdArgs &a, const RowsArgs *ra) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_act = a.seg_info[0], n_lo = a.seg_info[1];
    const int tu = (n_lo + H16 - 1) / H16;
    const int ti = (n_act - n_lo + H16 - 1) / H16;
    const int t = blockIdx.x;
    if (t >= tu + ti) return;
    const bool user = t < tu;
    const int r0 = user ? t * H16 : n_lo + (t - tu) * H16;
    const int r1 = min(r0 + H16, user ? n_lo : n_act);
    const int nrows = r1 - r0;
    const int side = user ? 0 : 1;
    const int C = (1 + a.n_mod) * HD, LDO = C + 4;
    constexpr int LDN = HD + 4;
    float *AN = lds;                                   // narrow rows [16][68]
    float *OutT = lds + a.out_off;                     // Out tile    [16][C + 4]
    __shared__ float s_c[H16];
    __shared__ int s_act[H16];
    if (tid < H16) {
        const int node = tid < nrows ? a.act[r0 + tid] : 0;
        s_act[tid] = node;
        const float cv = (tid < nrows && a.stage != 2) ? head_c(a, node) : 0.f;       // (stage 2 multiplies nothing by c)
        s_c[tid] = cv;
        if (a.c_out && tid < nrows && a.stage != 2) a.c_out[r0 + tid] = cv;
    }
    // out0 -> block 0 of the Out tile, narrow -> AN (compact rows: no index needed)
    if (a.stage != 1)
    for (int e = tid; e < H16 * (HD / 4); e += 256) {
        const int r = e / (HD / 4), c4 = e % (HD / 4);
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
        if (ROWS) {
            if (r < nrows) {
                rows_piece(*ra, (int64_t)a.act[r0 + r], c4, x, y);
                *reinterpret_cast<float4 *>(ra->narrow + (int64_t)(r0 + r) * ra->ld_narrow + 4 * c4) = y;      // (kept, as the rows launch does)
            }
        } else if (r < nrows) {
            int64_t col = 4 * c4;
            if (a.peer_dl) { const int q = (4 * c4) / a.peer_dl; col = q * a.peer_stride + (4 * c4 - q * a.peer_dl); }
            x = *reinterpret_cast<const float4 *>(a.out0 + (int64_t)(r0 + r) * a.ld_out0 + col);
            y = *reinterpret_cast<const float4 *>(a.narrow + (int64_t)(r0 + r) * a.ld_nar + col);
        }
        *reinterpret_cast<float4 *>(OutT + r * LDO + 4 * c4) = x;
        *reinterpret_cast<float4 *>(AN + r * LDN + 4 * c4) = y;
        if (r < nrows && (ROWS || a.out0 != a.OutAct))   // out0 handed over separately (or made here): block 0 of OutAct is written here
            *reinterpret_cast<float4 *>(a.OutAct + (int64_t)(r0 + r) * a.ld_out + 4 * c4) = x;
    }
    __syncthreads();                                   // s_act
    if (a.stage == 2) {
        // the feature blocks stage 1 left in OutAct (acc + c * b_m), + the shared part: the sum stage 0 forms in one expression
        const int per_row = a.n_mod * (HD / 4);
        for (int e = tid; e < H16 * per_row; e += 256) {
            const int r = e / per_row, q = e % per_row, m = q / (HD / 4), c4 = q % (HD / 4);
            float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
            float *dst = a.OutAct + (int64_t)(r0 + r) * a.ld_out + (m + 1) * HD + 4 * c4;
            if (r < nrows) pv = *reinterpret_cast<const float4 *>(dst);
            const float4 an = *reinterpret_cast<const float4 *>(AN + r * LDN + 4 * c4);
            const float4 v = make_float4(pv.x + an.x, pv.y + an.y, pv.z + an.z, pv.w + an.w);
            *reinterpret_cast<float4 *>(OutT + r * LDO + (m + 1) * HD + 4 * c4) = v;
            if (r < nrows) *reinterpret_cast<float4 *>(dst) = v;
        }
    }
    if (a.stage != 2)
    for (int m = 0; m < a.n_mod; ++m) {
        const int D4 = a.D[m] / 4, lda = a.D[m] + 4;
        float *Am = lds + a.a_off[m];
        for (int e = tid; e < H16 * D4; e += 256) {
            const int r = e / D4, c4 = e % D4;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nrows) {
                x = head_s4(a, m, (int64_t)s_act[r], c4);
                if (a.s_out) *reinterpret_cast<float4 *>(a.s_out + (int64_t)(r0 + r) * a.ld_sout + a.s_off[m] + 4 * c4) = x;
            }
            *reinterpret_cast<float4 *>(Am + r * lda + 4 * c4) = x;
        }
    }
    __syncthreads();
    const int ai = lane & 15, kq = lane >> 4;
    const int col = wave * 16 + ai;                    // this lane's output column inside a 64-column block
    // ---- stage 1: feature blocks
    if (a.stage != 2) {
#pragma unroll
    for (int m = 0; m < HMAXM; ++m) {
        if (m < a.n_mod) {
            const int K = a.D[m];
            const float *ap = lds + a.a_off[m] + ai * (K + 4) + kq;
            const float *bp = a.pk + a.off_Wm[m] + (int64_t)wave * (K / 4) * 64 + lane;
            const v4h acc = head16_run((v4h){0.f, 0.f, 0.f, 0.f}, ap, bp, K / 4);
            const float bm = a.bias_m[m] ? a.bias_m[m][col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kq + r;
                const float part = acc[r] + s_c[row] * bm;
                if (a.stage == 1) {
                    if (row < nrows) a.OutAct[(int64_t)(r0 + row) * a.ld_out + (m + 1) * HD + col] = part;
                    continue;
                }
                const float v = part + AN[row * LDN + col];
                OutT[row * LDO + (m + 1) * HD + col] = v;
                if (row < nrows) a.OutAct[(int64_t)(r0 + row) * a.ld_out + (m + 1) * HD + col] = v;
            }
        }
    }
    }
    if (a.stage == 1) return;
    __syncthreads();
    // ---- stage 2: fused Linear over the whole Out tile (K = C) and the single-modal heads (K = 64)
    {
        const float *ap = OutT + ai * LDO + kq;
        const float *bp = a.pk + a.off_Wf[side] + (int64_t)wave * (C / 4) * 64 + lane;
        const v4h acc = head16_run((v4h){0.f, 0.f, 0.f, 0.f}, ap, bp, C / 4);
        const float bb = a.bias_f[side] ? a.bias_f[side][col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            if (row < nrows) a.YAct[(int64_t)(r0 + row) * a.ld_y + col] = acc[r] + bb;
        }
    }
#pragma unroll
    for (int m = 0; m < HMAXM; ++m) {
        if (m < a.n_mod) {
            const float *ap = OutT + ai * LDO + (m + 1) * HD + kq;
            const float *bp = a.pk + a.off_Ws[m] + (int64_t)wave * (HD / 4) * 64 + lane;
            const v4h acc = head16_run((v4h){0.f, 0.f, 0.f, 0.f}, ap, bp, HD / 4);
            const float bb = a.bias_s[m] ? a.bias_s[m][col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kq + r;
                if (row < nrows) a.YAct[(int64_t)(r0 + row) * a.ld_y + (m + 1) * HD + col] = acc[r] + bb;
            }
        }
    }
}

__global__ __launch_bounds__(256) void head_fwd16_kernel(HeadFwdArgs a) { head_fwd16_body<false>(a, nullptr); }
__global__ __launch_bounds__(256) void head_rows_fwd16_kernel(HeadFwdArgs a, RowsArgs ra) { head_fwd16_body<true>(a, &ra); }

}  // namespace elimrec

using namespace elimrec;

extern "C" size_t elimrec_head_pack_floats(int n_mod, const int *D) {
    if (n_mod < 0 || n_mod > HMAXM) return 0;
    return (size_t)head_pack_layout(n_mod, D).total;
}

extern "C" size_t elimrec_head_pack_bwd_offset(int n_mod, const int *D) {
    if (n_mod < 0 || n_mod > HMAXM) return 0;
    return (size_t)head_pack_layout(n_mod, D).fwd_total;
}

static int head_fwd_fused_impl(int peer_world, int64_t peer_dl, const elimrec_head_src16 *src, const elimrec_head_rows *rows, const int32_t *d_act, const int32_t *d_seg_info, int64_t R, const float *d_out0,
                                      int64_t ld_out0, const float *d_narrow, int64_t ld_nar, const float *d_c, int n_mod,
                                      const float *const *d_S, const int64_t *ldS, const int *D, const float *const *d_Wm,
                                      const float *const *d_bm, const float *d_Wf_user, const float *d_bf_user,
                                      const float *d_Wf_item, const float *d_bf_item, const float *const *d_Ws,
                                      const float *const *d_bs, float *d_pack, size_t pack_floats, float *d_OutAct,
                                      int64_t ld_out, float *d_YAct, int64_t ld_y, int recdim, int phase, void *stream) {
    ELIMREC_REQUIRE(d_act && d_seg_info && d_out0 && d_narrow && (src || (d_c && d_S)) && d_Wm && d_Wf_user && d_Wf_item && d_Ws && d_pack &&
                        d_OutAct && d_YAct, "head_fwd_fused: null pointer");
    ELIMREC_REQUIRE(!src || (src->d_table && (src->dtype == 1 || src->dtype == 2) && src->row_elems % 8 == 0 &&
                             (!src->d_S_out || src->ld_S_out % 4 == 0)),
                    "head_fwd_fused_src16: a table of fp16 (1) / bf16 (2) rows, 16-byte aligned rows");
    if (recdim != HD || n_mod < 1 || n_mod > HMAXM) { set_error("head_fwd_fused: recdim must be %d and 1..%d feature tables", HD, HMAXM); return ELIMREC_E_UNSUPPORTED; }
    const int C = (1 + n_mod) * HD;
    ELIMREC_REQUIRE(ld_out0 % 4 == 0 && ld_nar % 4 == 0 && ld_out >= C && ld_y >= C, "head_fwd_fused: bad leading dimensions");
    ELIMREC_REQUIRE(pack_floats >= elimrec_head_pack_floats(n_mod, D), "head_fwd_fused: packed-weight buffer too small");
    ELIMREC_REQUIRE(phase >= 0 && phase <= 4, "head_fwd_fused: phase 0 (pack + head), 1 (pack only), 2 (head only), 3 / 4 (head in two launches)");
    if (R <= 0) return 0;
    HeadFwdArgs a = {};
    PackJobs pj = {};
    int64_t off = 0;
    int lds_f = H16 * (HD + 4);                      // narrow tile first
    int blocks = 0;
    auto add_job_g = [&](const float *W, int N, int K, int64_t sn, int64_t sk) {
        PackJob &j = pj.j[pj.n];
        j.W = W; j.ld = K; j.K = K; j.dst = off; j.N = N; j.sn = sn; j.sk = sk;
        pj.first_block[pj.n] = blocks;
        blocks += (N * K + 255) / 256 > 64 ? 64 : (N * K + 255) / 256;
        ++pj.n;
        const int64_t at = off;
        off += (int64_t)N * K;
        return at;
    };
    auto add_job = [&](const float *W, int K) { return add_job_g(W, HD, K, K, 1); };
    constexpr int rows_t = H16;
    for (int m = 0; m < n_mod; ++m) {
        ELIMREC_REQUIRE((src || (d_S[m] && ldS[m] % 4 == 0)) && d_Wm[m] && d_Ws[m] && D[m] > 0 && D[m] % 4 == 0, "head_fwd_fused: bad feature table %d", m);
        if (!src) { a.S[m] = d_S[m]; a.ldS[m] = ldS[m]; }
        a.s_off[m] = m == 0 ? 0 : a.s_off[m - 1] + D[m - 1];
        a.D[m] = D[m]; a.bias_m[m] = d_bm ? d_bm[m] : nullptr;
        a.a_off[m] = lds_f;
        lds_f += rows_t * (D[m] + 4);
        a.off_Wm[m] = add_job(d_Wm[m], D[m]);
    }
    a.off_Wf[0] = add_job(d_Wf_user, C);
    a.off_Wf[1] = add_job(d_Wf_item, C);
    for (int m = 0; m < n_mod; ++m) { a.off_Ws[m] = add_job(d_Ws[m], HD); a.bias_s[m] = d_bs ? d_bs[m] : nullptr; }
    // the head backward's operands B[k][c] = W[k][c] (common.h: head_pack_layout)
    add_job_g(d_Wf_user, C, HD, 1, C);
    add_job_g(d_Wf_item, C, HD, 1, C);
    for (int m = 0; m < n_mod; ++m) add_job_g(d_Ws[m], HD, HD, 1, HD);
    ELIMREC_REQUIRE(off == head_pack_layout(n_mod, D).total, "head_fwd_fused: pack layout mismatch");
    pj.first_block[pj.n] = blocks;
    a.out_off = lds_f; lds_f += rows_t * (C + 4);
    a.part_off = lds_f;
    if (phase == 4) {
        // the second launch of the two-launch head stages no feature rows: narrow tile + Out tile only (21 KB instead of 51 KB at
        // three 128-d tables: as many workgroups per CU as the registers allow -- a launch of thousands of tiles (large batches) is
        // paced by how many tiles a CU holds; at B = 2048 every tile is resident either way)
        a.out_off = H16 * (HD + 4);
        lds_f = a.out_off + rows_t * (C + 4);
    } else if (phase == 3) {
        // ... and the first launch stages nothing else: the feature tiles from offset 0
        int at = 0;
        for (int m = 0; m < n_mod; ++m) { a.a_off[m] = at; at += rows_t * (D[m] + 4); }
        lds_f = at;
    }
    const size_t lds_bytes = (size_t)lds_f * sizeof(float);
    if (lds_bytes > 158 * 1024) { set_error("head_fwd_fused: feature widths need %zu B of LDS", lds_bytes); return ELIMREC_E_UNSUPPORTED; }
    a.act = d_act; a.seg_info = d_seg_info; a.out0 = d_out0; a.ld_out0 = ld_out0; a.narrow = d_narrow; a.ld_nar = ld_nar; a.c = d_c;
    if (peer_world > 0) {
        ELIMREC_REQUIRE(!rows && peer_dl > 0 && peer_dl % 4 == 0 && peer_world * peer_dl == HD && ld_out0 == 2 * peer_dl && ld_nar == ld_out0,
                        "head_fwd_fused_peers: %d peers x %lld columns (a multiple of 4) must make the %d columns of a row", peer_world, (long long)peer_dl, HD);
        a.peer_dl = (int)peer_dl; a.peer_stride = R * 2 * peer_dl;
    }
    if (src) {
        a.sdtype = src->dtype; a.tab16 = (const uint16_t *)src->d_table; a.row_elems = src->row_elems;
        a.c_off = a.s_off[n_mod - 1] + D[n_mod - 1];
        ELIMREC_REQUIRE(a.c_off + 2 <= src->row_elems && a.c_off % 2 == 0, "head_fwd_fused_src16: rows shorter than sum(D) + 2 elements");
        a.s_out = src->d_S_out; a.ld_sout = src->ld_S_out; a.c_out = src->d_c_out;
    }
    a.n_mod = n_mod; a.pk = d_pack; a.bias_f[0] = d_bf_user; a.bias_f[1] = d_bf_item;
    a.OutAct = d_OutAct; a.ld_out = ld_out; a.YAct = d_YAct; a.ld_y = ld_y;
    hipStream_t s = (hipStream_t)stream;
    a.stage = phase == 3 ? 1 : (phase == 4 ? 2 : 0);
    if (phase < 2) {
        hipLaunchKernelGGL(pack_head_weights16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pj, d_pack);
        ELIMREC_LAUNCH_CHECK("pack_head_weights");
    }
    if (phase == 1) return 0;
    static size_t lds_set16 = 0;
    if (lds_bytes > 64 * 1024 && lds_bytes > lds_set16) {
        hipError_t e = hipFuncSetAttribute((const void *)head_fwd16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return check_hip(e, "head_fwd_fused: LDS size");
        lds_set16 = lds_bytes;
    }
    const unsigned tiles = (unsigned)((R + H16 - 1) / H16 + 2);     // user tiles + item tiles <= R/16 + 2
    if (rows) {
        RowsArgs ra = {};
        int rc = rows_args_fill("head_fwd_fused_rows", rows->A, rows->ns, rows->w, rows->L, rows->U, rows->layers, rows->d_long, ra);
        if (rc) return rc;
        ELIMREC_REQUIRE(ra.nc4 == HD / 4 && rows->d_narrow_out && rows->ld_narrow_out % 4 == 0 && phase == 4,
                        "head_fwd_fused_rows: %d table columns per row, a narrow buffer, phase 4", HD);
        ra.narrow = rows->d_narrow_out; ra.ld_narrow = rows->ld_narrow_out;
        static size_t lds_set16r = 0;
        if (lds_bytes > 64 * 1024 && lds_bytes > lds_set16r) {
            hipError_t e = hipFuncSetAttribute((const void *)head_rows_fwd16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return check_hip(e, "head_fwd_fused_rows: LDS size");
            lds_set16r = lds_bytes;
        }
        hipLaunchKernelGGL(head_rows_fwd16_kernel, dim3(tiles), dim3(256), lds_bytes, s, a, ra);
        ELIMREC_LAUNCH_CHECK("head_rows_fwd16");
        return 0;
    }
    hipLaunchKernelGGL(head_fwd16_kernel, dim3(tiles), dim3(256), lds_bytes, s, a);
    ELIMREC_LAUNCH_CHECK("head_fwd16");
    return 0;
}

extern "C" int elimrec_head_fwd_fused(const int32_t *d_act, const int32_t *d_seg_info, int64_t R, const float *d_out0,
                                      int64_t ld_out0, const float *d_narrow, int64_t ld_nar, const float *d_c, int n_mod,
                                      const float *const *d_S, const int64_t *ldS, const int *D, const float *const *d_Wm,
                                      const float *const *d_bm, const float *d_Wf_user, const float *d_bf_user,
                                      const float *d_Wf_item, const float *d_bf_item, const float *const *d_Ws,
                                      const float *const *d_bs, float *d_pack, size_t pack_floats, float *d_OutAct,
                                      int64_t ld_out, float *d_YAct, int64_t ld_y, int recdim, int phase, void *stream) {
    return head_fwd_fused_impl(0, 0, nullptr, nullptr, d_act, d_seg_info, R, d_out0, ld_out0, d_narrow, ld_nar, d_c, n_mod, d_S, ldS, D, d_Wm, d_bm,
                               d_Wf_user, d_bf_user, d_Wf_item, d_bf_item, d_Ws, d_bs, d_pack, pack_floats, d_OutAct, ld_out, d_YAct,
                               ld_y, recdim, phase, stream);
}

// ... with out0 / narrow read where the forward exchange of the column shards left them: d_recv = [world][R][out0 dl | narrow dl],
// peer q's columns of MY rows (ops.peer_cols_to_rows + elimrec_head_fwd_fused without the pass in between; same bits).
extern "C" int elimrec_head_fwd_fused_peers(const int32_t *d_act, const int32_t *d_seg_info, int64_t R, const float *d_recv, int world,
                                            int64_t dl, const float *d_c, int n_mod, const float *const *d_S, const int64_t *ldS,
                                            const int *D, const float *const *d_Wm, const float *const *d_bm, const float *d_Wf_user,
                                            const float *d_bf_user, const float *d_Wf_item, const float *d_bf_item,
                                            const float *const *d_Ws, const float *const *d_bs, float *d_pack, size_t pack_floats,
                                            float *d_OutAct, int64_t ld_out, float *d_YAct, int64_t ld_y, int recdim, int phase, void *stream) {
    ELIMREC_REQUIRE(d_recv && world > 0 && dl > 0 && (phase == 0 || phase == 2 || phase == 4), "head_fwd_fused_peers: a received buffer, phase 0, 2 or 4");
    return head_fwd_fused_impl(world, dl, nullptr, nullptr, d_act, d_seg_info, R, d_recv, 2 * dl, d_recv + dl, 2 * dl, d_c, n_mod, d_S, ldS, D,
                               d_Wm, d_bm, d_Wf_user, d_bf_user, d_Wf_item, d_bf_item, d_Ws, d_bs, d_pack, pack_floats, d_OutAct, ld_out,
                               d_YAct, ld_y, recdim, phase, stream);
}

// ... with the feature constants read from their 16-bit storage (lookup.hip's packed rows, one rank holding every row): see
// HeadFwdArgs::sdtype. Same phases; the same bits as elimrec_lookup_unpack(direct) followed by elimrec_head_fwd_fused on its rows.
extern "C" int elimrec_head_fwd_fused_src16(const elimrec_head_src16 *src, const int32_t *d_act, const int32_t *d_seg_info, int64_t R,
                                            const float *d_out0, int64_t ld_out0, const float *d_narrow, int64_t ld_nar, int n_mod,
                                            const int *D, const float *const *d_Wm, const float *const *d_bm, const float *d_Wf_user,
                                            const float *d_bf_user, const float *d_Wf_item, const float *d_bf_item,
                                            const float *const *d_Ws, const float *const *d_bs, float *d_pack, size_t pack_floats,
                                            float *d_OutAct, int64_t ld_out, float *d_YAct, int64_t ld_y, int recdim, int phase, void *stream) {
    ELIMREC_REQUIRE(src, "head_fwd_fused_src16: null source");
    return head_fwd_fused_impl(0, 0, src, nullptr, d_act, d_seg_info, R, d_out0, ld_out0, d_narrow, ld_nar, nullptr, n_mod, nullptr, nullptr, D, d_Wm,
                               d_bm, d_Wf_user, d_bf_user, d_Wf_item, d_bf_item, d_Ws, d_bs, d_pack, pack_floats, d_OutAct, ld_out, d_YAct,
                               ld_y, recdim, phase, stream);
}

extern "C" int elimrec_head_fwd_fused_rows(const elimrec_head_rows *rows, const int32_t *d_act, const int32_t *d_seg_info, int64_t R,
                                           const float *d_c, int n_mod, const float *const *d_S, const int64_t *ldS, const int *D,
                                           const float *const *d_Wm, const float *const *d_bm, const float *d_Wf_user,
                                           const float *d_bf_user, const float *d_Wf_item, const float *d_bf_item,
                                           const float *const *d_Ws, const float *const *d_bs, float *d_pack, size_t pack_floats,
                                           float *d_OutAct, int64_t ld_out, float *d_YAct, int64_t ld_y, int recdim, void *stream) {
    ELIMREC_REQUIRE(rows && d_OutAct, "head_fwd_fused_rows: null pointer");
    // (out0 / narrow of the plain entry are the buffers this launch fills itself: block 0 of OutAct and rows->d_narrow_out)
    return head_fwd_fused_impl(0, 0, nullptr, rows, d_act, d_seg_info, R, d_OutAct, ld_out, rows->d_narrow_out, rows->ld_narrow_out, d_c, n_mod, d_S, ldS, D,
                               d_Wm, d_bm, d_Wf_user, d_bf_user, d_Wf_item, d_bf_item, d_Ws, d_bs, d_pack, pack_floats, d_OutAct, ld_out,
                               d_YAct, ld_y, recdim, 4, stream);
}
