// Slab-major (column-sharded) LightGCN propagation: models/EliMRec.py:238-248 for a column slice of the folded
// d-column table (include/elimrec_hip.h, "slab-major propagation").
//
// Layout: a table of dl columns = ns slabs of width w floats, slab s = contiguous [n x w]. A hop is independent per
// column, so workgroups with equal blockIdx % gs work on the same slab group -- under the round-robin dispatch they
// share an XCD, whose 4 MiB L2 then has to hold [n x w*spg] floats (3.6 MB at the Tiktok shape for w*spg = 8) instead
// of the whole [n x d] table (29 MB) that a row-partitioned hop drags through every XCD's L2 (2.5x HBM over-fetch,
// profiles/r01_h_pmc_traffic.json). Placement changes speed only: every workgroup reads Xin and writes its own
// outputs, nothing is exchanged inside a launch.
//
// Work items (SELL-64, built on the host): rows with <= T non-zeros and <= T-nnz segments of longer rows, sorted by
// decreasing length; a lane group of LPR = spg*w/4 lanes owns one item, a wave 64/LPR items of similar length, and
// the (col, val) of step j of the wave's items are consecutive in memory. HBM/L2-bound (9 flop per 4-B gathered).
#include "common.h"
#include "merge_rows.h"
#include "rows_args.h"
#include "bwd_w.h"
#include <cstdlib>

namespace elimrec {


struct SellArgs {
    const int32_t *item_dst, *item_len, *blk_off, *col;
    const float *val;
    int64_t n_rows, n_src;
    int n_seg, n_long;
    int item_begin, item_end;      // items [begin, end), multiples of 64
    int seg_limit;                 // items below this index are segments of split rows
    int w4, w4_shift, gs, spg;
    const float4 *Xin;
    const uint32_t *src_mask;
    float4 *Xout;
    const float4 *Add;
    const uint32_t *add_mask;
    float scale;
    float4 *partials;
    const int32_t *long_rows, *long_seg_ptr;
    int compact_long;              // the fix-up writes Xout compactly: [ns x n_long x w]
};

typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool bit_of(const uint32_t *m, int r) { return (m[r >> 5] >> (r & 31)) & 1u; }

__device__ __forceinline__ void slab_epilogue(const SellArgs &a, int slab, int64_t row, int c4, float4 r) {
    const int64_t idx = ((int64_t)slab * a.n_rows + row) * a.w4 + c4;
    float4 s = r;
    if (a.Add && (!a.add_mask || bit_of(a.add_mask, (int)row))) {
        const float4 t = a.Add[idx];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    a.Xout[idx] = make_float4(s.x * a.scale, s.y * a.scale, s.z * a.scale, s.w * a.scale);
}

// sum_j val[j] * Xin[col[j]] in neighbour order with fmaf -- the arithmetic of half_gather (spmm.hip), so an unsplit
// row gets the same bits from either layout.
template <int LPR, bool MASKED, int U, bool PF>
__global__ __launch_bounds__(256) void sell_hop_kernel(SellArgs a) {
    constexpr int IPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int grp = (int)(blockIdx.x % (unsigned)a.gs);
    const int64_t wblk = (int64_t)(blockIdx.x / (unsigned)a.gs) * 4 + (threadIdx.x >> 6);
    const int64_t first = (int64_t)a.item_begin + wblk * IPW;
    if (first >= a.item_end) return;                               // wave-uniform
    const int64_t item = first + lane / LPR;
    const int cl = lane % LPR;
    const int len = a.item_len[item];
    const int dst = a.item_dst[item];
    const int slab = grp * a.spg + (cl >> a.w4_shift);
    const int c4 = cl & (a.w4 - 1);
    const float4 *X = a.Xin + (int64_t)slab * a.n_src * a.w4 + c4;
    const int64_t e0 = ((int64_t)a.blk_off[item >> 6] << 6) + (item & 63);
    const int32_t *colp = a.col + e0;
    const float *valp = a.val + e0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int ncj[U];
    float nvj[U];
    if (PF) {                      // the (col, val) of step j + U are in flight while step j gathers
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool in = u < len;
            ncj[u] = in ? colp[(int64_t)u << 6] : 0;
            nvj[u] = in ? valp[(int64_t)u << 6] : 0.f;
        }
    }
    for (int j = 0; j < len; j += U) {
        int cj[U];
        float vj[U];
        bool in[U];
        float4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            in[u] = (j + u) < len;
            if (PF) { cj[u] = ncj[u]; vj[u] = nvj[u]; }
            else {
                cj[u] = in[u] ? colp[(int64_t)(j + u) << 6] : 0;
                vj[u] = in[u] ? valp[(int64_t)(j + u) << 6] : 0.f;
            }
        }
        if (PF) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool nin = (j + U + u) < len;
                ncj[u] = nin ? colp[(int64_t)(j + U + u) << 6] : 0;
                nvj[u] = nin ? valp[(int64_t)(j + U + u) << 6] : 0.f;
            }
        }
        if (MASKED) {
#pragma unroll
            for (int u = 0; u < U; ++u) in[u] = in[u] && bit_of(a.src_mask, cj[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = in[u] ? X[(int64_t)cj[u] * a.w4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc.x = fmaf(vj[u], x[u].x, acc.x); acc.y = fmaf(vj[u], x[u].y, acc.y);
            acc.z = fmaf(vj[u], x[u].z, acc.z); acc.w = fmaf(vj[u], x[u].w, acc.w);
        }
    }
    if (dst < 0) return;                                           // padding item
    if (item < a.seg_limit) {
        a.partials[((int64_t)slab * a.n_seg + dst) * a.w4 + c4] = acc;
        return;
    }
    if (a.compact_long) return;                                    // seg_only launches carry no final items
    slab_epilogue(a, slab, dst, c4, acc);
}

// One wave per (split row, slab group): the wave's 64/LPR lane groups each add a contiguous share of the row's partial
// rows in order (8 loads in flight), the shares are added in group order through shuffles, group 0 writes the row.
template <int LPR>
__global__ __launch_bounds__(256) void sell_fixup_kernel(SellArgs a) {
    constexpr int NQ = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int grp = (int)(blockIdx.x % (unsigned)a.gs);
    const int li = (int)(blockIdx.x / (unsigned)a.gs) * 4 + (int)(threadIdx.x >> 6);
    if (li >= a.n_long) return;                                    // wave-uniform
    const int q = lane / LPR, cl = lane % LPR;
    const int slab = grp * a.spg + (cl >> a.w4_shift);
    const int c4 = cl & (a.w4 - 1);
    const int sb = a.long_seg_ptr[li], se = a.long_seg_ptr[li + 1];
    const int per = (se - sb + NQ - 1) / NQ;
    const int qb = min(sb + q * per, se), qe = min(qb + per, se);
    const float4 *P = a.partials + (int64_t)slab * a.n_seg * a.w4 + c4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int sgm = qb;
    for (; sgm + 8 <= qe; sgm += 8) {
        float4 p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = P[(int64_t)(sgm + u) * a.w4];
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += p[u].x; acc.y += p[u].y; acc.z += p[u].z; acc.w += p[u].w; }
    }
    for (; sgm < qe; ++sgm) {
        const float4 p = P[(int64_t)sgm * a.w4];
        acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
    }
    float4 tot = acc;
    if (NQ > 1) {
        tot.x = __shfl(acc.x, cl, 64); tot.y = __shfl(acc.y, cl, 64); tot.z = __shfl(acc.z, cl, 64); tot.w = __shfl(acc.w, cl, 64);
        const int nq = NQ + (a.n_long < 0 ? 1 : 0);      // = NQ; a run-time bound keeps the loop from being fully unrolled
#pragma unroll 2
        for (int t = 1; t < nq; ++t) {
            tot.x += __shfl(acc.x, t * LPR + cl, 64); tot.y += __shfl(acc.y, t * LPR + cl, 64);
            tot.z += __shfl(acc.z, t * LPR + cl, 64); tot.w += __shfl(acc.w, t * LPR + cl, 64);
        }
    }
    if (q != 0) return;
    if (a.compact_long) a.Xout[((int64_t)slab * a.n_long + li) * a.w4 + c4] = tot;
    else slab_epilogue(a, slab, a.long_rows[li], c4, tot);
}

// ---------------------------------------------------------------------------------------------------------------------
// Layer means at listed rows; hop L inline when its table is absent (rows_args.h: rows_piece). LR lanes per listed row, lane
// cl owns the float4 columns cl, cl + LR, ... of the row's nc4 = ns*w/4.
template <int LR>
__global__ __launch_bounds__(256) void slab_rows_kernel(RowsArgs a) {
    const int64_t s = (int64_t)blockIdx.x * (256 / LR) + threadIdx.x / LR;
    const int cl = threadIdx.x % LR;
    if (s >= a.R * a.n_lists) return;
    int64_t r = s;
    if (a.rows) {
        const int64_t list = s / a.R;
        if (a.counts && s - list * a.R >= a.counts[list]) return;
        r = a.rows[s];
        if (r < 0) return;                              // padding of a gathered list (negative keys); counts may be null
    }
    for (int c = cl; c < a.nc4; c += LR) {
        float4 out, nar;
        rows_piece(a, r, c, out, nar);
        *reinterpret_cast<float4 *>(a.out0 + s * a.ld_out0 + 4 * c) = out;
        *reinterpret_cast<float4 *>(a.narrow + (a.by_node ? r : s) * a.ld_narrow + 4 * c) = nar;
    }
}

__global__ void slab_from_rows_kernel(const float *__restrict__ src, int64_t ld, int64_t col0, int64_t n, int nc4, int w4,
                                      int w4_shift, float4 *__restrict__ slab) {
    const int64_t total = n * nc4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / nc4;
        const int c = (int)(e - r * nc4);
        slab[((int64_t)(c >> w4_shift) * n + r) * w4 + (c & (w4 - 1))] =
            *reinterpret_cast<const float4 *>(src + r * ld + col0 + 4 * c);
    }
}

__global__ void slab_to_rows_kernel(const float4 *__restrict__ slab, int64_t n, int nc4, int w4, int w4_shift,
                                    float *__restrict__ dst, int64_t ld, int64_t col0) {
    const int64_t total = n * nc4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / nc4;
        const int c = (int)(e - r * nc4);
        *reinterpret_cast<float4 *>(dst + r * ld + col0 + 4 * c) =
            slab[((int64_t)(c >> w4_shift) * n + r) * w4 + (c & (w4 - 1))];
    }
}

// Rank-ordered merge of [H | G] rows into the slab-major adjoint sources: a workgroup owns a range of node ids, finds
// the slice of every rank's ascending key list that falls into it (binary searches, one thread per rank) and walks the
// ranks IN RANK ORDER with a barrier in between; a node seen before (LDS bitmap) is accumulated, otherwise written.
__global__ __launch_bounds__(256) void slab_merge_rows_kernel(MergeArgs a) {
    __shared__ int s_beg[kSlabMaxRanks], s_end[kSlabMaxRanks];
    extern __shared__ uint32_t seen[];
    slab_merge_rows_body(a, (int)blockIdx.x, s_beg, s_end, seen);
}

__global__ void adam_out_kernel(const float *__restrict__ p_in, float *__restrict__ p_out, const float *__restrict__ g,
                                float *__restrict__ m, float *__restrict__ v, int64_t n, float step_size, float beta1,
                                float beta2, float inv_sqrt_bc2, float eps, float wd) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float pi = p_in[i];
        const float gi = fmaf(wd, pi, g[i]);
        const float mi = m[i] + (1.f - beta1) * (gi - m[i]);
        const float vi = fmaf(1.f - beta2, gi * gi, beta2 * v[i]);
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        m[i] = mi;
        v[i] = vi;
        p_out[i] = pi - step_size * (mi / denom);
    }
}

// =====================================================================================================================
// Argument block and lane helpers shared by the hop forms below. A lane owns VPL = 4 columns of a row piece (one 16-B load);
// split rows publish partial rows write-through (sc1), draw a ticket per row, and the wave that draws a row's last ticket adds
// the partials in segment order after ONE agent-scope acquire (cdna_hip_programming.md Guideline 16, recipe R1 in its counter
// form).
struct StreamArgs {
    const int32_t *item_dst, *item_len, *blk_off, *col, *item_long;
    const float *val;
    int64_t n_rows, n_src;
    int n_seg, n_long;
    int64_t n_blocks;              // wave blocks (64/LPR items each) to walk
    int seg_limit;
    int wl, wl_shift, gs, spg;     // wl = lanes per slab row
    const void *Xin;
    const uint32_t *src_mask;
    void *Xout;
    const float4 *Add;
    const uint32_t *add_mask;
    float scale;
    float *partials;
    const int32_t *long_rows, *long_seg_ptr;
    int32_t *tickets;              // [gs x n_long], zero between launches (self-resetting)
    int compact_long;
    // optional Adam epilogue (the adjoint's last hop, fp32 tables): the hop's output IS the gradient of the column shard, and
    // the optimizer update of a row piece is applied where it is produced instead of by a later pass that reads it back
    const float *ad_p_in;          // null: plain hop
    float *ad_p_out, *ad_m, *ad_v;
    float ad_step_size, ad_inv_sqrt_bc2, ad_beta1, ad_beta2, ad_eps, ad_wd;
    int ad_keep_grad;              // also write the gradient to Xout
};

template <int VPL, bool BF16>
__device__ __forceinline__ void lane_load(const void *base, int64_t idx, float (&x)[VPL]) {
    static_assert(VPL == 4 && !BF16, "fp32 tables: a lane owns four columns");
    const float4 t = ((const float4 *)base)[idx];
    x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
}

template <int VPL, bool BF16>
__device__ __forceinline__ void lane_store(void *base, int64_t idx, const float (&x)[VPL]) {
    static_assert(VPL == 4 && !BF16, "fp32 tables: a lane owns four columns");
#ifdef ELIMREC_NT_OUT
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v v = {x[0], x[1], x[2], x[3]};
    __builtin_nontemporal_store(v, ((f4v *)base) + idx);
#else
    ((float4 *)base)[idx] = make_float4(x[0], x[1], x[2], x[3]);
#endif
}

template <int VPL, bool OUT_BF16, bool ADAM = false>
__device__ __forceinline__ void stream_epilogue(const StreamArgs &a, int slab, int64_t row, int c, float (&r)[VPL]) {
    const int64_t idx = ((int64_t)slab * a.n_rows + row) * a.wl + c;
    if (a.Add && (!a.add_mask || bit_of(a.add_mask, (int)row))) {
        float t[VPL];
        lane_load<VPL, false>(a.Add, idx, t);
#pragma unroll
        for (int i = 0; i < VPL; ++i) r[i] += t[i];
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) r[i] *= a.scale;
    if (ADAM && VPL == 4 && !OUT_BF16) {               // arithmetic of adam_multi_kernel, element for element
#ifdef ELIMREC_NT_ADAM
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v p4 = __builtin_nontemporal_load((const f4v *)a.ad_p_in + idx), m4 = __builtin_nontemporal_load((const f4v *)a.ad_m + idx),
                  v4 = __builtin_nontemporal_load((const f4v *)a.ad_v + idx);
#else
        const float4 p4 = ((const float4 *)a.ad_p_in)[idx], m4 = ((const float4 *)a.ad_m)[idx], v4 = ((const float4 *)a.ad_v)[idx];
#endif
        const float pi[4] = {p4.x, p4.y, p4.z, p4.w}, mo[4] = {m4.x, m4.y, m4.z, m4.w}, vo[4] = {v4.x, v4.y, v4.z, v4.w};
        float po[4], mi[4], vi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float gi = fmaf(a.ad_wd, pi[i], r[i]);
            mi[i] = mo[i] + (1.f - a.ad_beta1) * (gi - mo[i]);
            vi[i] = fmaf(1.f - a.ad_beta2, gi * gi, a.ad_beta2 * vo[i]);
            const float denom = sqrtf(vi[i]) * a.ad_inv_sqrt_bc2 + a.ad_eps;
            po[i] = pi[i] - a.ad_step_size * (mi[i] / denom);
        }
#ifdef ELIMREC_NT_ADAM
        __builtin_nontemporal_store((f4v){mi[0], mi[1], mi[2], mi[3]}, (f4v *)a.ad_m + idx);
        __builtin_nontemporal_store((f4v){vi[0], vi[1], vi[2], vi[3]}, (f4v *)a.ad_v + idx);
#else
        ((float4 *)a.ad_m)[idx] = make_float4(mi[0], mi[1], mi[2], mi[3]);
        ((float4 *)a.ad_v)[idx] = make_float4(vi[0], vi[1], vi[2], vi[3]);
#endif
        ((float4 *)a.ad_p_out)[idx] = make_float4(po[0], po[1], po[2], po[3]);
        if (!a.ad_keep_grad) return;
    }
    lane_store<VPL, OUT_BF16>(a.Xout, idx, r);
}

// the whole wave on one split row (li wave-uniform): same order as sell_fixup_kernel
template <int LPR, int VPL, bool OUT_BF16, bool ADAM = false>
__device__ __forceinline__ void stream_combine(const StreamArgs &a, int grp, int li) {
    constexpr int NQ = 64 / LPR, UNR = VPL == 4 ? 8 : 4;
    const int lane = threadIdx.x & 63;
    const int q = lane / LPR, cl = lane % LPR;
    const int slab = grp * a.spg + (cl >> a.wl_shift);
    const int c = cl & (a.wl - 1);
    const int sb = a.long_seg_ptr[li], se = a.long_seg_ptr[li + 1];
    const int per = (se - sb + NQ - 1) / NQ;
    const int qb = min(sb + q * per, se), qe = min(qb + per, se);
    const int64_t pbase = (int64_t)slab * a.n_seg * a.wl + c;
    float acc[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) acc[i] = 0.f;
    int sgm = qb;
    for (; sgm + UNR <= qe; sgm += UNR) {
        float p[UNR][VPL];
#pragma unroll
        for (int u = 0; u < UNR; ++u) lane_load<VPL, false>(a.partials, pbase + (int64_t)(sgm + u) * a.wl, p[u]);
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int i = 0; i < VPL; ++i) acc[i] += p[u][i];
    }
    for (; sgm < qe; ++sgm) {
        float p[VPL];
        lane_load<VPL, false>(a.partials, pbase + (int64_t)sgm * a.wl, p);
#pragma unroll
        for (int i = 0; i < VPL; ++i) acc[i] += p[i];
    }
    float tot[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) tot[i] = NQ > 1 ? __shfl(acc[i], cl, 64) : acc[i];
    const int nq = NQ + (a.n_long < 0 ? 1 : 0);  // = NQ, but not a compile-time bound: an unrolled loop keeps every
#pragma unroll 2                                  // shuffle result (a register each) in flight -- 256 VGPRs at 32 groups
    for (int g = 1; g < nq; ++g) {              // group order
#pragma unroll
        for (int i = 0; i < VPL; ++i) tot[i] += __shfl(acc[i], g * LPR + cl, 64);
    }
    if (q != 0) return;
    if (a.compact_long) lane_store<VPL, false>(a.Xout, ((int64_t)slab * a.n_long + li) * a.wl + c, tot);
    else stream_epilogue<VPL, OUT_BF16, ADAM>(a, slab, a.long_rows[li], c, tot);
}

// The whole optimizer step in ONE launch: up to 8 jobs (the column shard of the embeddings, read from one buffer and
// written to the other; the spans of the projection weights that have a
// gradient, in place; copy-only spans). A job may also copy its PRE-update parameters to copy_dst: the snapshot of the
// projection weights the cached tables were computed with (models/EliMRec.py:98-99). Arithmetic of adam_kernel.
// The same body also runs as extra workgroups at the end of the hop + Adam launch (sell_tier_kernel<..., ADAM>).
struct AdamJob {
    const float *p_in;
    float *p_out;
    const float *g;         // nullable: copy-only job
    float *m, *v;
    float *copy_dst;        // nullable
    int64_t n;
    float step_size, inv_sqrt_bc2;
    int first_block;
};
struct AdamJobs {
    AdamJob j[8]; int n; int blocks;
    const float *sum_src; float *sum_dst; int sum_n;      // one more workgroup: sum_dst[0] = sum of sum_src[0 .. sum_n) in elimrec_sum's order
};

// elimrec_sum (bpr.hip sum_kernel) by one workgroup of 256 threads: 1024 virtual threads add x[t], x[t + 1024], ..., then the
// binary tree over the 1024 partials -- the same additions, so the same bits
__device__ __forceinline__ void fixed_order_sum_body(const float *__restrict__ x, int n, float *__restrict__ out, float *s) {
    for (int vt = threadIdx.x; vt < 1024; vt += 256) {
        float acc = 0.f;
        for (int i = vt; i < n; i += 1024) acc += x[i];
        s[vt] = acc;
    }
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        for (int idx = threadIdx.x; idx < w; idx += 256) s[idx] += s[idx + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s[0];
}

__device__ __forceinline__ void adam_jobs_body(const AdamJobs &jobs, int block, float beta1, float beta2, float eps, float wd) {
    int k = 0;
    while (k + 1 < jobs.n && block >= jobs.j[k + 1].first_block) ++k;
    const AdamJob &jb = jobs.j[k];
    const int nb = (k + 1 < jobs.n ? jobs.j[k + 1].first_block : jobs.blocks) - jb.first_block;
    for (int64_t i = (int64_t)(block - jb.first_block) * 256 + threadIdx.x; i < jb.n; i += (int64_t)nb * 256) {
        const float pi = jb.p_in[i];
        if (jb.copy_dst) jb.copy_dst[i] = pi;
        if (!jb.g) continue;
        const float gi = fmaf(wd, pi, jb.g[i]);
        const float mi = jb.m[i] + (1.f - beta1) * (gi - jb.m[i]);
        const float vi = fmaf(1.f - beta2, gi * gi, beta2 * jb.v[i]);
        const float denom = sqrtf(vi) * jb.inv_sqrt_bc2 + eps;
        jb.m[i] = mi;
        jb.v[i] = vi;
        const float po = pi - jb.step_size * (mi / denom);
        jb.p_out[i] = po;
    }
}

__global__ __launch_bounds__(256) void adam_multi_kernel(AdamJobs jobs, float beta1, float beta2, float eps, float wd) {
    adam_jobs_body(jobs, (int)blockIdx.x, beta1, beta2, eps, wd);
}

// ---------------------------------------------------------------------------------------------------------------------
// Form 3 (whole fp32 tables; the engine's default there): ONE launch per hop over WAVE TILES. What bounds the older
// forms at the Tiktok shape is not HBM but the CU's memory pipeline walking the index: every (col, val) pair costs two
// wave-wide load instructions that return 32 useful bytes each, a step of 4 neighbours depends on the index loads of
// the step before, and a hop that gathers nothing still takes 28 of the 34 us (tools/bench_tier.py). Here a tile is
// the work of one wave -- G = 64/LPR lane groups -- with its index stored [step][group]: the wave reads it with
// full-width coalesced loads (lane l takes entry l, l+64, ...; two loads ahead), and a lane group picks its neighbour
// of step j out of the loaded registers with a cross-lane read (ds_bpermute), so the index costs one load instruction
// per 64 neighbours instead of one per G, and the chain is index -> gathers with 8 gathers per lane in flight.
// Tiles by row length (T = long_threshold, G groups):
//   * rows of <= T non-zeros: G rows per tile, a lane group each, SELL order (sum in neighbour order, = CSR order);
//   * T < length <= T1: one wave per row, the row's neighbours dealt round-robin to the G groups, the group sums added
//     in group order through shuffles;
//   * T1 < length <= T2: one workgroup per row -- four such tiles over contiguous quarters, the four wave sums added in
//     wave order through LDS;
//   * longer (a dozen rows at the Tiktok shape): T-long segments, G per tile; a segment wave publishes its partial rows
//     write-through, draws a ticket per row, and the wave that draws a row's last ticket adds the partials in segment
//     order after ONE agent-scope acquire (cdna_hip_programming.md Guideline 16, recipe R1 in its counter form).
// Every sum has a fixed order, so results are bitwise reproducible; rows above T differ in round-off from forms 0-2.
struct TierArgs {
    StreamArgs s;
    const int64_t *tile_off;       // first index entry of a tile: 64-bit (>= 2^31 entries at configs[4])
    const int32_t *tile_len, *tile_dst, *tile_long, *tcol, *long_index;
    const float *tval;
    int n_w4;                      // workgroup rows (4 tiles each)
    int t1_base, tseg_base, tfin_base, n_tiles;     // first tile of the wave rows, segment tiles, unsplit-row tiles
    int n_tiles_run;               // tiles this launch walks (seg_only: up to tfin_base)
    const uint64_t *ballots;       // masked hop: [n_tiles x kmax] source-mask bits of each 64-entry index line
    int kmax;
};

// The first adjoint hop gathers from a row-sparse table (<= 3B active rows). Testing the row bitmap per neighbour would
// cost a scattered load instruction per 64/LPR neighbours -- as many memory instructions as the gathers of a full hop,
// on the pipeline that bounds it -- so the bitmap is looked up ONCE per index entry by this pass (coalesced index read,
// L1-resident bitmap) and the hop kernel reads one 64-bit word per index line through the scalar cache.
__global__ __launch_bounds__(256) void tile_ballot_kernel(const int64_t *__restrict__ tile_off, const int32_t *__restrict__ tcol,
                                                          const uint32_t *__restrict__ mask, int G, int n_tiles, int kmax,
                                                          uint64_t *__restrict__ ballots) {
    // a wave per (tile, index line): blockIdx.y = line, so every wave is one short independent chain
    const int lane = threadIdx.x & 63;
    const int ti = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    const int k = (int)blockIdx.y;
    if (ti >= n_tiles) return;
    const int64_t off = tile_off[ti];
    const int nk = (int)((tile_off[ti + 1] - off + 63) >> 6);
    if (k >= nk) return;
    const int cidx = tcol[off + (k << 6) + lane];
    const uint64_t b = __ballot(bit_of(mask, cidx));
    if (lane == 0) ballots[(int64_t)ti * kmax + k] = b;
}

// -DELIMREC_NT_INDEX / -DELIMREC_NT_OUT: non-temporal index loads / output stores (measured at the Tiktok shape: index
// 32.4 -> 35.5 us per hop, output 32.4 -> 31.6 with the masked hop 0.8 us slower; the build sets ELIMREC_NT_OUT and, for the Adam
// epilogue's once-per-step streams, ELIMREC_NT_ADAM: see the Makefile)
#ifdef ELIMREC_NT_INDEX
#define ELIMREC_IDX_LD(p) __builtin_nontemporal_load(p)
#else
#define ELIMREC_IDX_LD(p) (*(p))
#endif
// one batch of UB neighbours per lane group: predicate, gather, fused multiply-adds in neighbour order
template <int VPL, bool IN_BF16, int UB>
__device__ __forceinline__ void tile_batch(const StreamArgs &a, int64_t in_base, const int (&cj)[UB], const float (&vj)[UB],
                                           const bool (&in)[UB], float (&acc)[VPL]) {
    float x[UB][VPL];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        if (in[u]) lane_load<VPL, IN_BF16>(a.Xin, in_base + (int64_t)cj[u] * a.wl, x[u]);
        else {
#pragma unroll
            for (int i = 0; i < VPL; ++i) x[u][i] = 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u)
#pragma unroll
        for (int i = 0; i < VPL; ++i) acc[i] = fmaf(vj[u], x[u][i], acc[i]);
}

// sum over the tile's steps of val * Xin[col] for this lane group; neighbour order, fmaf. An index line (64 entries)
// holds LPR steps of the G groups; a batch is UB steps: a part of a line (LPR >= UB) or UB/LPR whole lines (narrow
// row pieces: column shards).
template <int LPR, int VPL, bool IN_BF16, bool MASKED>
__device__ __forceinline__ void tile_gather(const TierArgs &t, int ti, int64_t in_base, int64_t off, int steps, int glen, int lane, int sub,
                                            float (&acc)[VPL]) {
    constexpr int G = 64 / LPR;
#ifdef ELIMREC_TILE_UB
    constexpr int UB = ELIMREC_TILE_UB;
#else
    constexpr int UB = 8;                                  // neighbours per lane group in flight
#endif
    const StreamArgs &a = t.s;
#pragma unroll
    for (int i = 0; i < VPL; ++i) acc[i] = 0.f;
    const int nk = (steps * G + 63) >> 6;                  // index lines
    const int32_t *colp = t.tcol + off + lane;
    const float *valp = t.tval + off + lane;
    if constexpr (LPR >= UB) {
        int c0 = 0, c1 = 0;
        float v0 = 0.f, v1 = 0.f;
        // MASKED (the first adjoint hop: <= 3B active source rows, a few per cent of the index entries): the source bits of an
        // index line are ONE 64-bit word read through the scalar cache, three lines ahead of its use -- a line none of whose entries
        // is active (most of them) costs neither its two index loads nor the cross-lane picks and predicated gathers of its steps
        const uint64_t *balp = MASKED ? t.ballots + (int64_t)ti * t.kmax : nullptr;
        uint64_t b0 = ~0ull, b1 = ~0ull, b2 = ~0ull;
        if (MASKED) {
            b0 = nk > 0 ? balp[0] : 0ull; b1 = nk > 1 ? balp[1] : 0ull; b2 = nk > 2 ? balp[2] : 0ull;
        }
        if (nk > 0 && b0) { c0 = ELIMREC_IDX_LD(colp); v0 = ELIMREC_IDX_LD(valp); }
        if (nk > 1 && b1) { c1 = ELIMREC_IDX_LD(colp + 64); v1 = ELIMREC_IDX_LD(valp + 64); }
        for (int k = 0; k < nk; ++k) {
            int c2 = 0;
            float v2 = 0.f;
            if (k + 2 < nk && b2) { c2 = ELIMREC_IDX_LD(colp + ((k + 2) << 6)); v2 = ELIMREC_IDX_LD(valp + ((k + 2) << 6)); }
            const uint64_t b3 = (MASKED && k + 3 < nk) ? balp[k + 3] : (MASKED ? 0ull : ~0ull);
            const int jbase = k * LPR;
            const uint64_t bal = b0;
            if (MASKED && bal == 0ull) {                   // wave-uniform
                c0 = c1; v0 = v1; c1 = c2; v1 = v2; b0 = b1; b1 = b2; b2 = b3;
                continue;
            }
#pragma unroll
            for (int u0 = 0; u0 < LPR; u0 += UB) {
                if (jbase + u0 >= steps) continue;         // wave-uniform
                int cj[UB];
                float vj[UB];
                bool in[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int srcl = (u0 + u) * G + sub;
                    cj[u] = __shfl(c0, srcl, 64);
                    vj[u] = __shfl(v0, srcl, 64);
                    in[u] = (jbase + u0 + u) < glen;
                    if (MASKED) in[u] = in[u] && ((bal >> srcl) & 1ull);
                }
                tile_batch<VPL, IN_BF16, UB>(a, in_base, cj, vj, in, acc);
            }
            c0 = c1; v0 = v1; c1 = c2; v1 = v2; b0 = b1; b1 = b2; b2 = b3;
        }
    } else {
        constexpr int KL = UB / LPR;                       // lines per batch
        int cc[KL], cn[KL];
        float vc[KL], vn[KL];
#pragma unroll
        for (int q = 0; q < KL; ++q) {
            cc[q] = q < nk ? colp[q << 6] : 0;
            vc[q] = q < nk ? valp[q << 6] : 0.f;
        }
        for (int k = 0; k < nk; k += KL) {
#pragma unroll
            for (int q = 0; q < KL; ++q) {
                const bool more = k + KL + q < nk;
                cn[q] = more ? colp[(k + KL + q) << 6] : 0;
                vn[q] = more ? valp[(k + KL + q) << 6] : 0.f;
            }
            const int jbase = k * LPR;
            uint64_t bal[KL];
#pragma unroll
            for (int q = 0; q < KL; ++q) bal[q] = (MASKED && k + q < nk) ? t.ballots[(int64_t)ti * t.kmax + k + q] : 0ull;
            int cj[UB];
            float vj[UB];
            bool in[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int q = u / LPR, srcl = (u % LPR) * G + sub;
                cj[u] = G == 64 ? cc[q] : __shfl(cc[q], srcl, 64);
                vj[u] = G == 64 ? vc[q] : __shfl(vc[q], srcl, 64);
                in[u] = (jbase + u) < glen;
                if (MASKED) in[u] = in[u] && ((bal[q] >> srcl) & 1ull);
            }
            tile_batch<VPL, IN_BF16, UB>(a, in_base, cj, vj, in, acc);
#pragma unroll
            for (int q = 0; q < KL; ++q) { cc[q] = cn[q]; vc[q] = vn[q]; }
        }
    }
}

// the lane groups' sums of one wave added in group order; valid in the lanes of group 0
template <int LPR, int VPL>
__device__ __forceinline__ void tier_wave_sum(const float (&acc)[VPL], int cl, int n_long_guard, float (&tot)[VPL]) {
    constexpr int NQ = 64 / LPR;
#pragma unroll
    for (int i = 0; i < VPL; ++i) tot[i] = NQ > 1 ? __shfl(acc[i], cl, 64) : acc[i];
    const int nq = NQ + (n_long_guard < 0 ? 1 : 0);          // run-time bound: see stream_combine
#pragma unroll 2
    for (int g = 1; g < nq; ++g) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) tot[i] += __shfl(acc[i], g * LPR + cl, 64);
    }
}

template <int LPR, int VPL, bool IN_BF16, bool OUT_BF16, bool MASKED, bool ADAM, bool EXT_LDS = false>
__device__ __forceinline__ void tier_body(const TierArgs &t, float *lds = nullptr, unsigned front = 0) {
    constexpr int G = 64 / LPR;
    const StreamArgs &a = t.s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned bx = blockIdx.x - front;                 // (front: a multiple of 8 -- the workgroup's XCD and slab group keep their relation)
    const int grp = (int)(bx % (unsigned)a.gs);            // workgroup b works on slab group b % gs: an XCD keeps to one group
    const int bidx = (int)(bx / (unsigned)a.gs);
    const int sub = lane / LPR, cl = lane % LPR;
    const int slab = grp * a.spg + (cl >> a.wl_shift);
    const int c = cl & (a.wl - 1);
    const int64_t in_base = (int64_t)slab * a.n_src * a.wl + c;
    __shared__ float s_own[EXT_LDS ? 1 : 4 * 64 * 8];
    float *s_part = EXT_LDS ? lds : s_own;                  // 4 * 64 * 8 floats
    const int ti = __builtin_amdgcn_readfirstlane(bidx * 4 + wave);       // tiles are laid out workgroup by workgroup
    const bool wg_row = bidx < t.n_w4;
    int64_t off = 0;
    int steps = 0, glen = 0, dst = -1;
    if (ti < t.n_tiles_run) {
        off = t.tile_off[ti];
        steps = (int)((t.tile_off[ti + 1] - off) / G);
        glen = t.tile_len[(int64_t)ti * G + sub];
        dst = t.tile_dst[(int64_t)ti * G + sub];
    }
    if (a.compact_long && a.add_mask && ti < t.n_tiles_run) {
        // the split rows only (seg_only), and of those only the WANTED ones (a.add_mask: bit r = row r's sum is read -- the rows of the
        // batch): a row nobody reads costs no gathers. A workgroup's / a wave's row is the same for all its lanes; a segment tile's
        // lane groups belong to different rows -- one that is not wanted publishes nothing and draws no ticket (all segments of a
        // row decide alike, so its tickets stay at rest)
        if (wg_row || ti < t.tseg_base) {
            const int row = __shfl(dst, 0, 64);
            if (row >= 0 && !bit_of(a.add_mask, row)) return;          // (workgroup rows: the four waves share the row -- all return)
        } else if (dst >= 0) {
            const int li0 = t.tile_long[(int64_t)(ti - t.tseg_base) * G + sub];
            if (!bit_of(a.add_mask, a.long_rows[li0])) { dst = -1; glen = 0; }
        }
    }
    float acc[VPL];
    tile_gather<LPR, VPL, IN_BF16, MASKED>(t, ti, in_base, off, steps, glen, lane, sub, acc);
    if (wg_row) {
        // ---- a workgroup per row: wave sums through LDS, wave order
        float tot[VPL];
        tier_wave_sum<LPR, VPL>(acc, cl, a.n_long, tot);
        if (sub == 0) {
#pragma unroll
            for (int i = 0; i < VPL; ++i) s_part[(wave * LPR + cl) * VPL + i] = tot[i];
        }
        __syncthreads();
        if (wave == 0 && sub == 0) {
#pragma unroll
            for (int i = 0; i < VPL; ++i) tot[i] = ((s_part[cl * VPL + i] + s_part[(LPR + cl) * VPL + i]) + s_part[(2 * LPR + cl) * VPL + i]) +
                                                   s_part[(3 * LPR + cl) * VPL + i];
            if (a.compact_long) lane_store<VPL, false>(a.Xout, ((int64_t)slab * a.n_long + t.long_index[dst]) * a.wl + c, tot);
            else stream_epilogue<VPL, OUT_BF16, ADAM>(a, slab, dst, c, tot);
        }
        return;
    }
    if (ti >= t.n_tiles_run) return;
    if (ti < t.tseg_base) {
        // ---- a wave per row
        const int row = __shfl(dst, 0, 64);
        if (row < 0) return;                               // padding tile
        float tot[VPL];
        tier_wave_sum<LPR, VPL>(acc, cl, a.n_long, tot);
        if (sub == 0) {
            if (a.compact_long) lane_store<VPL, false>(a.Xout, ((int64_t)slab * a.n_long + t.long_index[row]) * a.wl + c, tot);
            else stream_epilogue<VPL, OUT_BF16, ADAM>(a, slab, row, c, tot);
        }
        return;
    }
    if (ti >= t.tfin_base) {
        // ---- unsplit rows, a lane group each
        if (dst >= 0) stream_epilogue<VPL, OUT_BF16, ADAM>(a, slab, dst, c, acc);
        return;
    }
    // ---- segments of the longest rows: partial rows, tickets, last arriver combines
    __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)a.partials, 0, (int)min((size_t)0x7FFFFFF0, (size_t)a.n_seg * a.gs * a.spg * a.wl * VPL * 4), 0x00020000);
    if (dst >= 0) {
        const int64_t pidx = ((int64_t)slab * a.n_seg + dst) * a.wl + c;
#pragma unroll
        for (int h = 0; h < VPL / 4; ++h) {
            u32x4s bits = {__float_as_uint(acc[4 * h]), __float_as_uint(acc[4 * h + 1]), __float_as_uint(acc[4 * h + 2]),
                           __float_as_uint(acc[4 * h + 3])};
            __builtin_amdgcn_raw_buffer_store_b128(bits, prsrc, (unsigned)((pidx * (VPL / 4) + h) * 16), 0, 16 /* sc1 */);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int li = -1, ticket = -1, nseg = 0;
    if (dst >= 0) {
        li = t.tile_long[(int64_t)(ti - t.tseg_base) * G + sub];
        nseg = a.long_seg_ptr[li + 1] - a.long_seg_ptr[li];
        if (cl == 0) ticket = __hip_atomic_fetch_add(&a.tickets[(int64_t)grp * a.n_long + li], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ticket = __shfl(ticket, sub * LPR, 64);
    const bool last = dst >= 0 && ticket == nseg - 1;
    if (__ballot(last) == 0ull) return;                               // wave-uniform
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        if (!__shfl(last ? 1 : 0, g * LPR, 64)) continue;
        const int g_li = __shfl(li, g * LPR, 64);
        stream_combine<LPR, VPL, OUT_BF16, ADAM>(a, grp, g_li);
        if (lane == 0) __hip_atomic_store(&a.tickets[(int64_t)grp * a.n_long + g_li], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// (a launch bound of five waves per SIMD changes no register count -- the kernel takes 98 VGPRs either way -- but the schedule
// the compiler picks under it runs the plain hop in 29.5 us against 31.7 and the step in 0.305 ms against 0.307, same bits)
#ifndef ELIMREC_TILE_WAVES
#define ELIMREC_TILE_WAVES 5
#endif
template <int LPR, int VPL, bool IN_BF16, bool OUT_BF16, bool MASKED>
__global__ __launch_bounds__(256, ELIMREC_TILE_WAVES) void sell_tier_kernel(TierArgs t) {
    tier_body<LPR, VPL, IN_BF16, OUT_BF16, MASKED, false>(t);
}

// the adjoint's last hop: Adam of the shard as the epilogue, and the projection weights' optimizer spans (which need
// nothing of this hop) as extra workgroups behind the tiles -- its own kernel, so that the other hops' argument block stays small
template <int LPR>
__global__ __launch_bounds__(256) void sell_tier_adam_kernel(TierArgs t, AdamJobs tail, int tail_block0) {
    __shared__ float scratch[4 * 64 * 8];                   // the hop's wave sums; the loss sum's 1024 partials
    if ((int)blockIdx.x >= tail_block0) {
        const int b = (int)blockIdx.x - tail_block0;
        if (b < tail.blocks) adam_jobs_body(tail, b, t.s.ad_beta1, t.s.ad_beta2, t.s.ad_eps, t.s.ad_wd);
        else fixed_order_sum_body(tail.sum_src, tail.sum_n, tail.sum_dst, scratch);     // the step's loss (needed by the host only)
        return;
    }
    tier_body<LPR, 4, false, false, false, true, true>(t, scratch);
}

// An adjoint hop with a phase of the projections' weight gradients (bwd_w.h) as extra workgroups behind the tiles: the
// partial launch (tail_mode 2; it needs the head backward's rows, as the adjoint's first hop does) or the fixed-order slab
// reduce (tail_mode 1; needs the partial launch, is needed by the optimizer only). Neither reads what the hop writes.
// Kernels of their own for the reason above: the plain hops keep their small argument block.
template <int LPR, bool MASKED>
__global__ __launch_bounds__(256, (LPR <= 8 ? 5 : 1)) void sell_tier_bwdw_kernel(TierArgs t, BwdBatch batch, int tail_block0, int tail_mode, int gx) {
    __shared__ float As[2][TRB * TN1];                      // the GEMM stages; the hop's wave sums (8 KB) share them
    __shared__ float Bs[2][TRB * TN2];
    static_assert(2 * TRB * TN1 >= 4 * 64 * 8, "the hop's LDS scratch must fit the A stages");
    // tail_block0 < 0: the tail's -tail_block0 workgroups (rounded up to a multiple of 8) come FIRST
    const int front = tail_block0 < 0 ? (-tail_block0 + 7) & ~7 : 0;
    if (tail_block0 < 0 ? (int)blockIdx.x < front : (int)blockIdx.x >= tail_block0) {
        const int b = tail_block0 < 0 ? (int)blockIdx.x : (int)blockIdx.x - tail_block0;
        if (tail_block0 < 0 && b >= -tail_block0) return;
        if (tail_mode == 1) { reduce_slabs_body(batch, b % gx, b / gx); return; }
        bwd_w_partial_body(batch, b, As, Bs);
        return;
    }
    tier_body<LPR, 4, false, false, MASKED, false, true>(t, &As[0][0], (unsigned)front);
}

static int log2_pow2(int x) {
    int s = 0;
    while ((1 << s) < x) ++s;
    return (1 << s) == x ? s : -1;
}

}  // namespace elimrec

using namespace elimrec;

static size_t slab_partial_floats_bytes(const elimrec_sell *A, int ns, int w) {
    return align_up((size_t)(A->n_seg > 0 ? A->n_seg : 1) * ns * w * sizeof(float), 256);
}
static size_t slab_ticket_bytes(const elimrec_sell *A) { return align_up((size_t)8 * (A->n_long > 0 ? A->n_long : 1) * sizeof(int32_t), 256); }
static size_t slab_ballot_bytes(const elimrec_sell *A) {
    if (!A->tiered) return 0;
    const size_t tiles = (size_t)A->n_t4 + A->n_t1 + A->n_tseg + A->n_tfin;
    return align_up(tiles * (size_t)(A->tile_kmax > 0 ? A->tile_kmax : 1) * sizeof(uint64_t), 256);
}

// partial rows + the arrival counters of the in-launch combine (8 slab groups at most); the caller zeroes the buffer ONCE
extern "C" size_t elimrec_slab_partials_bytes(const elimrec_sell *A, int ns, int w) {
    if (!A || ns <= 0 || w <= 0) return 0;
    return slab_partial_floats_bytes(A, ns, w) + slab_ticket_bytes(A) + slab_ballot_bytes(A);
}

struct AdamEpilogue {
    const float *p_in; float *p_out, *m, *v;
    float step_size, inv_sqrt_bc2, beta1, beta2, eps, wd;
    int keep_grad;
    const AdamJobs *tail;          // nullable
};

static int launch_tier(const elimrec_sell *A, int ns, int wl, int wl_shift, int gs, int spg, int lpr, const void *Xin,
                       const uint32_t *src_mask, void *Xout, const float *add,
                       const uint32_t *add_mask, float scale, float *partials, int flags, hipStream_t s,
                       const AdamEpilogue *adam = nullptr, const BwdBatch *bwdw = nullptr, int tail_mode = 0, int tail_blocks = 0,
                       int reduce_gx = 1) {
    const int seg_only = flags & 1;
    const bool bits_ready = (flags & 2) != 0;          // elimrec_slab_source_bits has run for this source bitmap
    if (A->tile_groups != 64 / lpr) {
        set_error("slab_hop: the plan's wave tiles were laid out for %d lane groups per wave, this table geometry has %d",
                  A->tile_groups, 64 / lpr);
        return ELIMREC_E_BADARG;
    }
    ELIMREC_REQUIRE(A->d_tile_off && A->d_tile_len && A->d_tile_dst && A->d_tile_col && A->d_tile_val && A->n_t4 == 4 * A->n_w4 &&
                        A->n_t1 % 4 == 0 && A->n_tseg % 4 == 0 && A->n_tfin % 4 == 0 && (A->n_tseg == 0 || A->d_tile_long),
                    "slab_hop: bad tile plan");
    TierArgs t = {};
    StreamArgs &a = t.s;
    a.n_rows = A->n_rows; a.n_src = A->n_src; a.n_seg = A->n_seg; a.n_long = A->n_long;
    a.wl = wl; a.wl_shift = wl_shift; a.gs = gs; a.spg = spg;
    a.Xin = Xin; a.src_mask = src_mask; a.Xout = Xout;
    a.Add = seg_only ? nullptr : (const float4 *)add; a.add_mask = add_mask; a.scale = seg_only ? 1.0f : scale;
    a.partials = partials; a.long_rows = A->d_long_rows; a.long_seg_ptr = A->d_long_seg_ptr;
    a.tickets = (int32_t *)((char *)partials + slab_partial_floats_bytes(A, ns, wl * 4));
    a.compact_long = seg_only ? 1 : 0;
    if (adam) {
        a.ad_p_in = adam->p_in; a.ad_p_out = adam->p_out; a.ad_m = adam->m; a.ad_v = adam->v;
        a.ad_step_size = adam->step_size; a.ad_inv_sqrt_bc2 = adam->inv_sqrt_bc2; a.ad_beta1 = adam->beta1; a.ad_beta2 = adam->beta2;
        a.ad_eps = adam->eps; a.ad_wd = adam->wd; a.ad_keep_grad = adam->keep_grad;
    }
    t.tile_off = A->d_tile_off; t.tile_len = A->d_tile_len; t.tile_dst = A->d_tile_dst; t.tile_long = A->d_tile_long;
    t.tcol = A->d_tile_col; t.tval = A->d_tile_val; t.long_index = A->d_long_index;
    t.n_w4 = A->n_w4;
    t.t1_base = A->n_t4; t.tseg_base = t.t1_base + A->n_t1; t.tfin_base = t.tseg_base + A->n_tseg;
    t.n_tiles = t.tfin_base + A->n_tfin;
    t.n_tiles_run = seg_only ? t.tfin_base : t.n_tiles;
    const int64_t per_group = t.n_tiles_run / 4;
    if (per_group <= 0) return 0;
    // (the slab groups one after the other instead of side by side -- what the chip gathers from at any time being ONE group's
    // slice -- measured at the configs[3] shape: 6.60 against 6.51 ms per step; the hop is bound by the CUs' gather rate, not by where
    // the lines come from. Removed.)
    int tail_block0 = (int)(per_group * gs);
    {
        // the weight gradients' partial launch AHEAD of the hop's tiles instead of behind them: it is a serial chain per workgroup
        // (18.9 us as a launch of its own), and begun at once it ends under the tiles -- measured -1 us per step at B = 2048, -5 at
        // 4096, -15 at 32768; the slab reduce first: +-0 (it stays behind). Same bits either way.
        if (bwdw && tail_blocks > 0 && !adam && tail_mode == 2) {
            tail_block0 = -tail_blocks;
            tail_blocks = (tail_blocks + 7) & ~7;
        }
    }
    AdamJobs tail = {};
    if (adam && adam->tail && (adam->tail->n > 0 || adam->tail->sum_src)) tail = *adam->tail;
    if (tail.n <= 0) tail.blocks = 0;
    const dim3 grid((unsigned)(per_group * gs)), grid_adam((unsigned)(per_group * gs) + (unsigned)tail.blocks + (tail.sum_src ? 1u : 0u));
    const bool masked = src_mask != nullptr;
    if (masked) {
        ELIMREC_REQUIRE(A->tile_kmax > 0, "slab_hop: bad tile plan (tile_kmax)");
        t.kmax = A->tile_kmax;
        uint64_t *ballots = (uint64_t *)((char *)a.tickets + slab_ticket_bytes(A));
        t.ballots = ballots;
        if (!bits_ready)
        hipLaunchKernelGGL(tile_ballot_kernel, dim3((unsigned)per_group, (unsigned)t.kmax), dim3(256), 0, s, t.tile_off, t.tcol, src_mask, 64 / lpr,
                           t.n_tiles_run, t.kmax, ballots);
    }
#define ELIMREC_TIER(LPR)                                                                                                     \
    do {                                                                                                                      \
        if (adam) hipLaunchKernelGGL((sell_tier_adam_kernel<LPR>), grid_adam, dim3(256), 0, s, t, tail, tail_block0);         \
        else if (bwdw && masked)                                                                                              \
            hipLaunchKernelGGL((sell_tier_bwdw_kernel<LPR, true>), dim3(grid.x + (unsigned)tail_blocks), dim3(256), 0, s, t,  \
                               *bwdw, tail_block0, tail_mode, reduce_gx);                                                     \
        else if (bwdw)                                                                                                        \
            hipLaunchKernelGGL((sell_tier_bwdw_kernel<LPR, false>), dim3(grid.x + (unsigned)tail_blocks), dim3(256), 0, s, t, \
                               *bwdw, tail_block0, tail_mode, reduce_gx);                                                     \
        else if (masked) hipLaunchKernelGGL((sell_tier_kernel<LPR, 4, false, false, true>), grid, dim3(256), 0, s, t);        \
        else hipLaunchKernelGGL((sell_tier_kernel<LPR, 4, false, false, false>), grid, dim3(256), 0, s, t);                   \
    } while (0)
    switch (lpr) {
        case 1: ELIMREC_TIER(1); break;
        case 2: ELIMREC_TIER(2); break;
        case 4: ELIMREC_TIER(4); break;
        case 8: ELIMREC_TIER(8); break;
        case 16: ELIMREC_TIER(16); break;
        case 32: ELIMREC_TIER(32); break;
        default: ELIMREC_TIER(64); break;
    }
#undef ELIMREC_TIER
    return check_hip(hipGetLastError(), "slab_hop(tiered)");
}

static int slab_geometry(const char *who, int ns, int w, int gs, int &w4_shift, int &spg, int &lpr) {
    const int w4 = w / 4;
    w4_shift = (w > 0 && w % 4 == 0) ? log2_pow2(w4) : -1;
    if (ns < 1 || w4_shift < 0) { set_error("%s: slab width must be 4 * 2^k (got w=%d, ns=%d)", who, w, ns); return ELIMREC_E_BADARG; }
    if (gs < 1 || ns % gs != 0) { set_error("%s: %d slab groups do not divide %d slabs", who, gs, ns); return ELIMREC_E_BADARG; }
    spg = ns / gs;
    if (log2_pow2(spg) < 0 || spg * w4 > 64) {
        set_error("%s: slabs per group (%d) must be a power of two with spg*w/4 <= 64", who, spg);
        return ELIMREC_E_BADARG;
    }
    lpr = spg * w4;
    return 0;
}

static int slab_simple_geometry(const char *who, int64_t n, int ns, int w, int &w4_shift) {
    w4_shift = (w > 0 && w % 4 == 0) ? log2_pow2(w / 4) : -1;
    if (n < 0 || ns < 1 || w4_shift < 0) { set_error("%s: bad slab geometry (n=%lld, ns=%d, w=%d)", who, (long long)n, ns, w); return ELIMREC_E_BADARG; }
    return 0;
}

extern "C" int elimrec_slab_hop(const elimrec_sell *A, int ns, int w, int gs, const float *d_Xin,
                                const uint32_t *d_src_mask, float *d_Xout, const float *d_add,
                                const uint32_t *d_add_mask, float scale, float *d_partials, size_t partials_bytes,
                                int flags, void *stream) {
    const int seg_only = flags & 1;
    ELIMREC_REQUIRE(A && d_Xin && d_Xout, "slab_hop: null pointer");
    ELIMREC_REQUIRE(d_Xin != d_Xout, "slab_hop: Xout must not alias Xin");
    ELIMREC_REQUIRE(A->n_items % 64 == 0 && A->n_seg_items % 64 == 0 && A->n_seg_items <= A->n_items, "slab_hop: bad plan");
    int w4_shift, spg, lpr, rc;
    if ((rc = slab_geometry("slab_hop", ns, w, gs, w4_shift, spg, lpr))) return rc;
    if ((A->n_long > 0 || A->tiered) && (!d_partials || partials_bytes < elimrec_slab_partials_bytes(A, ns, w))) {
        set_error("slab_hop: partial-row scratch too small");
        return ELIMREC_E_WORKSPACE;
    }
    SellArgs a = {};
    a.item_dst = A->d_item_dst; a.item_len = A->d_item_len; a.blk_off = A->d_blk_off; a.col = A->d_col; a.val = A->d_val;
    a.n_rows = A->n_rows; a.n_src = A->n_src; a.n_seg = A->n_seg; a.n_long = A->n_long;
    a.item_begin = 0; a.item_end = seg_only ? A->n_seg_items : A->n_items; a.seg_limit = A->n_seg_items;
    a.w4 = w / 4; a.w4_shift = w4_shift; a.gs = gs; a.spg = spg;
    a.Xin = (const float4 *)d_Xin; a.src_mask = d_src_mask; a.Xout = (float4 *)d_Xout;
    a.Add = seg_only ? nullptr : (const float4 *)d_add; a.add_mask = d_add_mask; a.scale = seg_only ? 1.0f : scale;
    a.partials = (float4 *)d_partials; a.long_rows = A->d_long_rows; a.long_seg_ptr = A->d_long_seg_ptr;
    a.compact_long = seg_only ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (A->tiered)
        return launch_tier(A, ns, w / 4, w4_shift, gs, spg, lpr, d_Xin, d_src_mask, d_Xout, d_add, d_add_mask, scale,
                           d_partials, flags, s);
    const int n_it = a.item_end - a.item_begin;
    if (n_it > 0) {
        const int ipw = 64 / lpr;
        const unsigned blocks = (unsigned)(((int64_t)n_it / ipw + 3) / 4) * (unsigned)gs;
#define ELIMREC_SELL_LAUNCH(LPR)                                                                          \
    do {                                                                                                  \
        if (d_src_mask) hipLaunchKernelGGL((sell_hop_kernel<LPR, true, 8, false>), dim3(blocks), dim3(256), 0, s, a); \
        else hipLaunchKernelGGL((sell_hop_kernel<LPR, false, 4, true>), dim3(blocks), dim3(256), 0, s, a);            \
    } while (0)
        switch (lpr) {
            case 1: ELIMREC_SELL_LAUNCH(1); break;
            case 2: ELIMREC_SELL_LAUNCH(2); break;
            case 4: ELIMREC_SELL_LAUNCH(4); break;
            case 8: ELIMREC_SELL_LAUNCH(8); break;
            case 16: ELIMREC_SELL_LAUNCH(16); break;
            case 32: ELIMREC_SELL_LAUNCH(32); break;
            default: ELIMREC_SELL_LAUNCH(64); break;
        }
#undef ELIMREC_SELL_LAUNCH
        ELIMREC_LAUNCH_CHECK("slab_hop");
    }
    if (A->n_long > 0) {
        const unsigned blocks = (unsigned)((A->n_long + 3) / 4) * (unsigned)gs;
        switch (lpr) {
            case 1: hipLaunchKernelGGL((sell_fixup_kernel<1>), dim3(blocks), dim3(256), 0, s, a); break;
            case 2: hipLaunchKernelGGL((sell_fixup_kernel<2>), dim3(blocks), dim3(256), 0, s, a); break;
            case 4: hipLaunchKernelGGL((sell_fixup_kernel<4>), dim3(blocks), dim3(256), 0, s, a); break;
            case 8: hipLaunchKernelGGL((sell_fixup_kernel<8>), dim3(blocks), dim3(256), 0, s, a); break;
            case 16: hipLaunchKernelGGL((sell_fixup_kernel<16>), dim3(blocks), dim3(256), 0, s, a); break;
            case 32: hipLaunchKernelGGL((sell_fixup_kernel<32>), dim3(blocks), dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL((sell_fixup_kernel<64>), dim3(blocks), dim3(256), 0, s, a); break;
        }
        ELIMREC_LAUNCH_CHECK("slab_hop(fixup)");
    }
    return 0;
}

static int build_adam_jobs(const char *who, const elimrec_adam_job *jobs, int n_jobs, float lr, float beta1, float beta2, AdamJobs &a) {
    a.n = 0;
    int blocks = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const elimrec_adam_job &j = jobs[k];
        if (j.n <= 0) continue;
        ELIMREC_REQUIRE(j.d_p_in, "%s: job %d has no parameters", who, k);
        ELIMREC_REQUIRE(!j.d_g || (j.d_p_out && j.d_m && j.d_v && j.step >= 1), "%s: job %d: an update needs p_out, m, v and a 1-based step", who, k);
        ELIMREC_REQUIRE(j.d_g || j.d_copy_dst, "%s: job %d does nothing", who, k);
        AdamJob &o = a.j[a.n++];
        o.p_in = j.d_p_in; o.p_out = j.d_p_out; o.g = j.d_g; o.m = j.d_m; o.v = j.d_v;
        o.copy_dst = j.d_copy_dst; o.n = j.n;
        if (j.d_g) {
            const double bc1 = 1.0 - pow((double)beta1, (double)j.step);
            const double bc2 = 1.0 - pow((double)beta2, (double)j.step);
            o.step_size = (float)((double)lr / bc1);
            o.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
        }
        o.first_block = blocks;
        int64_t nb = (j.n + 255) / 256;
        if (nb > 4096) nb = 4096;
        blocks += (int)nb;
    }
    a.blocks = blocks;
    return 0;
}

extern "C" int elimrec_slab_source_bits(const elimrec_sell *A, int ns, int w, int gs, const uint32_t *d_src_mask,
                                        float *d_partials, size_t partials_bytes, void *stream) {
    ELIMREC_REQUIRE(A && d_src_mask && d_partials, "slab_source_bits: null pointer");
    ELIMREC_REQUIRE(A->tiered && A->tile_kmax > 0, "slab_source_bits: needs a tiered (wave-tile) plan");
    int w4_shift, spg, lpr, rc;
    if ((rc = slab_geometry("slab_source_bits", ns, w, gs, w4_shift, spg, lpr))) return rc;
    ELIMREC_REQUIRE(A->tile_groups == 64 / lpr, "slab_source_bits: the plan's tiles were laid out for another geometry");
    if (partials_bytes < elimrec_slab_partials_bytes(A, ns, w)) { set_error("slab_source_bits: scratch too small"); return ELIMREC_E_WORKSPACE; }
    const int n_tiles = A->n_t4 + A->n_t1 + A->n_tseg + A->n_tfin;
    if (n_tiles <= 0) return 0;
    uint64_t *ballots = (uint64_t *)((char *)d_partials + slab_partial_floats_bytes(A, ns, w) + slab_ticket_bytes(A));
    hipLaunchKernelGGL(tile_ballot_kernel, dim3((unsigned)(n_tiles / 4), (unsigned)A->tile_kmax), dim3(256), 0, (hipStream_t)stream,
                       A->d_tile_off, A->d_tile_col, d_src_mask, 64 / lpr, n_tiles, A->tile_kmax, ballots);
    ELIMREC_LAUNCH_CHECK("slab_source_bits");
    return 0;
}

extern "C" int elimrec_slab_hop_bwd_w(const elimrec_sell *A, int ns, int w, int gs, const float *d_Xin, const uint32_t *d_src_mask,
                                      float *d_Xout, const float *d_add, const uint32_t *d_add_mask, float scale, float *d_partials,
                                      size_t partials_bytes, int flags, const elimrec_linear_bwd_desc *descs, int n,
                                      void *d_workspace, size_t workspace_bytes, int phase, void *stream) {
    ELIMREC_REQUIRE(A && d_Xin && d_Xout && d_Xin != d_Xout, "slab_hop_bwd_w: null pointer / alias");
    ELIMREC_REQUIRE(A->tiered && !(flags & 1), "slab_hop_bwd_w: needs a tiered (wave-tile) plan and a full hop");
    ELIMREC_REQUIRE(phase == 0 || phase == 1, "slab_hop_bwd_w: phase 0 (partial launch) or 1 (slab reduce)");
    ELIMREC_REQUIRE(descs && n >= 1 && n <= kMaxBatch, "slab_hop_bwd_w: 1..%d problems", kMaxBatch);
    ELIMREC_REQUIRE(d_workspace && workspace_bytes >= bwd_w_batched_bytes(descs, n), "slab_hop_bwd_w: weight-gradient workspace");
    int w4_shift, spg, lpr, rc;
    if ((rc = slab_geometry("slab_hop_bwd_w", ns, w, gs, w4_shift, spg, lpr))) return rc;
    if (!d_partials || partials_bytes < elimrec_slab_partials_bytes(A, ns, w)) {
        set_error("slab_hop_bwd_w: partial-row scratch too small");
        return ELIMREC_E_WORKSPACE;
    }
    BwdBatch batch;
    int blocks = 0, max_out = 0;
    if ((rc = bwd_w_build_batch(descs, n, d_workspace, batch, blocks, max_out))) return rc;
    const int gx = (4 * max_out + 255) / 256;
    return launch_tier(A, ns, w / 4, w4_shift, gs, spg, lpr, d_Xin, d_src_mask, d_Xout, d_add, d_add_mask, scale,
                       d_partials, flags, (hipStream_t)stream, nullptr, &batch, phase == 0 ? 2 : 1, phase == 0 ? blocks : gx * n, gx);
}

extern "C" int elimrec_slab_hop_adam(const elimrec_sell *A, int ns, int w, int gs, const float *d_Xin, float *d_grad_out,
                                     const float *d_add, const uint32_t *d_add_mask, float scale, float *d_partials,
                                     size_t partials_bytes, const float *d_p_in, float *d_p_out, float *d_m, float *d_v, float lr,
                                     float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                     const elimrec_adam_job *tail_jobs, int n_tail_jobs, const float *d_sum_src, int64_t sum_n,
                                     float *d_sum_dst, void *stream) {
    ELIMREC_REQUIRE(A && d_Xin && d_p_in && d_p_out && d_m && d_v, "slab_hop_adam: null pointer");
    ELIMREC_REQUIRE(!d_sum_src || (d_sum_dst && sum_n >= 0 && sum_n < INT32_MAX), "slab_hop_adam: the sum needs a destination");
    ELIMREC_REQUIRE(A->tiered, "slab_hop_adam: needs a tiered (wave-tile) plan");
    ELIMREC_REQUIRE(step >= 1, "slab_hop_adam: 1-based step");
    ELIMREC_REQUIRE((const void *)d_Xin != (const void *)d_p_out && (const void *)d_Xin != (const void *)d_m &&
                        (const void *)d_Xin != (const void *)d_v, "slab_hop_adam: the gathered table must not be written");
    int w4_shift, spg, lpr, rc;
    if ((rc = slab_geometry("slab_hop_adam", ns, w, gs, w4_shift, spg, lpr))) return rc;
    if (!d_partials || partials_bytes < elimrec_slab_partials_bytes(A, ns, w)) {
        set_error("slab_hop_adam: partial-row scratch too small");
        return ELIMREC_E_WORKSPACE;
    }
    AdamEpilogue ad;
    ad.p_in = d_p_in; ad.p_out = d_p_out; ad.m = d_m; ad.v = d_v;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    ad.step_size = (float)((double)lr / bc1);
    ad.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    ad.beta1 = beta1; ad.beta2 = beta2; ad.eps = eps; ad.wd = weight_decay; ad.keep_grad = d_grad_out ? 1 : 0;
    AdamJobs tail = {};
    ad.tail = nullptr;
    if (tail_jobs && n_tail_jobs > 0) {
        ELIMREC_REQUIRE(n_tail_jobs <= 8, "slab_hop_adam: at most 8 tail jobs");
        if ((rc = build_adam_jobs("slab_hop_adam", tail_jobs, n_tail_jobs, lr, beta1, beta2, tail))) return rc;
        ad.tail = &tail;
    }
    if (d_sum_src) { tail.sum_src = d_sum_src; tail.sum_dst = d_sum_dst; tail.sum_n = (int)sum_n; ad.tail = &tail; }
    return launch_tier(A, ns, w / 4, w4_shift, gs, spg, lpr, d_Xin, nullptr, d_grad_out, d_add, d_add_mask, scale,
                       d_partials, 0, (hipStream_t)stream, &ad);
}

extern "C" int elimrec_slab_rows(const elimrec_sell *A, int ns, int w, int L, int64_t U, const float *const *layers,
                                 const float *d_long, const int32_t *d_rows, const int32_t *d_counts, int64_t R,
                                 int n_lists, float *d_out0, int64_t ld_out0, float *d_narrow, int64_t ld_narrow,
                                 int narrow_by_node, void *stream) {
    ELIMREC_REQUIRE(A && layers && d_out0 && d_narrow, "slab_rows: null pointer");
    ELIMREC_REQUIRE(ld_out0 % 4 == 0 && ld_narrow % 4 == 0, "slab_rows: leading dimensions must be multiples of 4");
    ELIMREC_REQUIRE(n_lists >= 1 && (d_rows || n_lists == 1), "slab_rows: several lists need row ids");
    RowsArgs a = {};
    int rc;
    if ((rc = rows_args_fill("slab_rows", A, ns, w, L, U, layers, d_long, a))) return rc;
    a.rows = d_rows; a.counts = d_counts; a.R = R; a.n_lists = n_lists;
    a.out0 = d_out0; a.ld_out0 = ld_out0; a.narrow = d_narrow; a.ld_narrow = ld_narrow; a.by_node = narrow_by_node;
    const int64_t total = R * n_lists;
    if (total <= 0) return 0;
    int lr = 1;
    while (lr < a.nc4 && lr < 64) lr *= 2;
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((total + (256 / lr) - 1) / (256 / lr));
    switch (lr) {
        case 1: hipLaunchKernelGGL((slab_rows_kernel<1>), dim3(blocks), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((slab_rows_kernel<2>), dim3(blocks), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((slab_rows_kernel<4>), dim3(blocks), dim3(256), 0, s, a); break;
        case 8: hipLaunchKernelGGL((slab_rows_kernel<8>), dim3(blocks), dim3(256), 0, s, a); break;
        case 16: hipLaunchKernelGGL((slab_rows_kernel<16>), dim3(blocks), dim3(256), 0, s, a); break;
        case 32: hipLaunchKernelGGL((slab_rows_kernel<32>), dim3(blocks), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((slab_rows_kernel<64>), dim3(blocks), dim3(256), 0, s, a); break;
    }
    ELIMREC_LAUNCH_CHECK("slab_rows");
    return 0;
}

extern "C" int elimrec_slab_from_rows(const float *d_src, int64_t ld, int64_t col0, int64_t n, int ns, int w,
                                      float *d_slab, void *stream) {
    ELIMREC_REQUIRE(d_src && d_slab && ld % 4 == 0 && col0 % 4 == 0, "slab_from_rows: bad arguments");
    int sh, rc;
    if ((rc = slab_simple_geometry("slab_from_rows", n, ns, w, sh))) return rc;
    if (n == 0) return 0;
    int64_t blocks = (n * ns * (w / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(slab_from_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_src, ld, col0, n,
                       ns * (w / 4), w / 4, sh, (float4 *)d_slab);
    ELIMREC_LAUNCH_CHECK("slab_from_rows");
    return 0;
}

extern "C" int elimrec_slab_to_rows(const float *d_slab, int64_t n, int ns, int w, float *d_dst, int64_t ld, int64_t col0,
                                    void *stream) {
    ELIMREC_REQUIRE(d_dst && d_slab && ld % 4 == 0 && col0 % 4 == 0, "slab_to_rows: bad arguments");
    int sh, rc;
    if ((rc = slab_simple_geometry("slab_to_rows", n, ns, w, sh))) return rc;
    if (n == 0) return 0;
    int64_t blocks = (n * ns * (w / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(slab_to_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)d_slab,
                       n, ns * (w / 4), w / 4, sh, d_dst, ld, col0);
    ELIMREC_LAUNCH_CHECK("slab_to_rows");
    return 0;
}

extern "C" int elimrec_slab_merge_rows(const float *d_rows, const int32_t *d_keys, int world, int64_t R, int64_t U,
                                       int64_t I, int ns, int w, int M, float *d_SrcA, float *d_SrcB, uint32_t *d_mask,
                                       void *stream) {
    ELIMREC_REQUIRE(M >= -1, "slab_merge_rows: M >= 0, or -1 for [H | G] rows without the user / item side swap");
    const int plain = M < 0 ? 1 : 0;
    if (M < 0) M = 0;
    ELIMREC_REQUIRE(d_rows && d_keys && d_SrcA && d_SrcB && d_mask, "slab_merge_rows: null pointer");
    ELIMREC_REQUIRE(world >= 1 && world <= kSlabMaxRanks && R >= 1 && R < INT32_MAX, "slab_merge_rows: 1..%d ranks", kSlabMaxRanks);
    int sh, rc;
    const int64_t N = U + I;
    if ((rc = slab_simple_geometry("slab_merge_rows", N, ns, w, sh))) return rc;
    MergeArgs a = {d_rows, d_keys, world, (int)R, U, N, ns * (w / 4), w / 4, sh, merge_rows_chunk(N), M, d_SrcA, d_SrcB, d_mask, plain};
    const unsigned grid = (unsigned)((N + a.chunk - 1) / a.chunk);
    if (grid == 0) return 0;
    hipLaunchKernelGGL(slab_merge_rows_kernel, dim3(grid), dim3(256), (size_t)(a.chunk / 32) * sizeof(uint32_t), (hipStream_t)stream, a);
    ELIMREC_LAUNCH_CHECK("slab_merge_rows");
    return 0;
}

// Row bitmap of `world` sorted key lists (negative padding behind the valid prefix): bit n <=> node n is in some list. The
// words elimrec_slab_merge_rows writes for the same lists -- available as soon as the ids are gathered, i.e. a whole forward
// pass before the rows themselves arrive, so the masked hop's source bits can be prepared off the critical path.
namespace elimrec {
__global__ __launch_bounds__(256) void rows_bitmap_kernel(const int32_t *__restrict__ keys, int W, int R, int64_t N, int chunk,
                                                          uint32_t *__restrict__ mask) {
    extern __shared__ uint32_t seen[];
    __shared__ int s_beg[kSlabMaxRanks], s_end[kSlabMaxRanks];
    const int tid = threadIdx.x;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(lo + chunk, N);
    for (int w = tid; w < chunk / 32; w += 256) seen[w] = 0u;
    if (tid < W) {
        const int32_t *kr = keys + (int64_t)tid * R;
        int a0 = 0, a1 = R, b0 = 0, b1 = R;
        while (a0 < a1 || b0 < b1) {
            if (a0 < a1) { const int m = (a0 + a1) >> 1; if (slab_merge_key(kr, m) < lo) a0 = m + 1; else a1 = m; }
            if (b0 < b1) { const int m = (b0 + b1) >> 1; if (slab_merge_key(kr, m) < hi) b0 = m + 1; else b1 = m; }
        }
        s_beg[tid] = a0; s_end[tid] = b0;
    }
    __syncthreads();
    for (int r = 0; r < W; ++r)
        for (int s = s_beg[r] + tid; s < s_end[r]; s += 256) {
            const int bit = (int)((int64_t)keys[(int64_t)r * R + s] - lo);
            atomicOr(&seen[bit >> 5], 1u << (bit & 31));
        }
    __syncthreads();
    for (int w = tid; w < chunk / 32; w += 256)
        if (lo + 32 * (int64_t)w < ((N + 31) / 32) * 32) mask[lo / 32 + w] = seen[w];
}
}  // namespace elimrec

extern "C" int elimrec_rows_bitmap(const int32_t *d_keys, int world, int64_t R, int64_t N, uint32_t *d_mask, void *stream) {
    ELIMREC_REQUIRE(d_keys && d_mask && world >= 1 && world <= kSlabMaxRanks && R >= 1 && R < INT32_MAX && N >= 1, "rows_bitmap: bad arguments");
    const int chunk = merge_rows_chunk(N);
    hipLaunchKernelGGL(rows_bitmap_kernel, dim3((unsigned)((N + chunk - 1) / chunk)), dim3(256), (size_t)(chunk / 32) * sizeof(uint32_t),
                       (hipStream_t)stream, d_keys, world, (int)R, N, chunk, d_mask);
    ELIMREC_LAUNCH_CHECK("rows_bitmap");
    return 0;
}

extern "C" int elimrec_adam_step_out(const float *d_p_in, float *d_p_out, const float *d_g, float *d_m, float *d_v,
                                     int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                     int64_t step, void *stream) {
    ELIMREC_REQUIRE(d_p_in && d_p_out && d_g && d_m && d_v, "adam_step_out: null pointer");
    ELIMREC_REQUIRE(step >= 1, "adam_step_out: step is 1-based");
    if (n <= 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_out_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_p_in, d_p_out, d_g, d_m,
                       d_v, n, step_size, beta1, beta2, inv_sqrt_bc2, eps, weight_decay);
    ELIMREC_LAUNCH_CHECK("adam_step_out");
    return 0;
}

extern "C" int elimrec_adam_multi(const elimrec_adam_job *jobs, int n_jobs, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, void *stream) {
    ELIMREC_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= 8, "adam_multi: 1..8 jobs");
    AdamJobs a = {};
    int rc = build_adam_jobs("adam_multi", jobs, n_jobs, lr, beta1, beta2, a);
    if (rc) return rc;
    if (a.n == 0) return 0;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)a.blocks), dim3(256), 0, (hipStream_t)stream, a, beta1, beta2, eps, weight_decay);
    ELIMREC_LAUNCH_CHECK("adam_multi");
    return 0;
}
