// One LightGCN hop for the rows of ONE side of the bipartite graph whose sources are a table far larger than the caches:
//     Xout[r] = (sum_j A[r, j] Xin[j] + [add_mask bit r] Add[r]) * scale      for the swept rows r in [0, n_sweep)
// (models/EliMRec.py:243-247 for one column slice; the user rows of BASELINE.json configs[3] / configs[4], which gather from
// 1.2 M / 100 M item rows -- 623 MB per 128-column table at configs[3]).
//
// The tile hop (slab.hip) lets every workgroup walk its rows' neighbour lists from end to end, so at any moment the gathers
// of a launch are spread over the whole source table: every source row piece is fetched from beyond L2 once per
// neighbour (7.4 times each at configs[3]: 5.8 GB past L2 for a 0.64 GB table, profiles/r03_c4_hop_traffic.json). Here the
// SOURCE range is swept in windows that fit an XCD's 4 MB L2, by all workgroups of the XCD together:
//   * an XCD owns one slab (128-B row pieces of one column slice) and 1 / parts of the swept rows; each of its workgroups
//     keeps the OUTPUT row pieces of one block of rows in LDS for the whole sweep;
//   * the rows of a block are dealt to the workgroup's waves (about equal non-zero counts per wave), and the plan
//     (slab.SweepPlan) lays a wave's non-zeros out window by window, row by row inside a window; the entries of one (wave,
//     window) are cut at row boundaries into one chunk per lane group of the wave, and every chunk into 80-byte step records of
//     eight entries -- the same number of records for every lane group of the (wave, window), empty ones where a chunk is
//     shorter. The wave's stream is then a plain sequence of STEPS, a record per lane group each, contiguous in memory: the
//     records of the next steps are loaded while this step's eight pieces per lane are gathered (no chunk bounds, no window
//     bookkeeping in the kernel -- the windows are in the order of the stream). Each gathered piece is added to its row's
//     accumulator, read from LDS with the step's other rows' and written back when the step leaves the row. A row never leaves
//     its wave, so nothing synchronises inside the sweep;
//   * blocks and waves carry about equal numbers of non-zeros and every window the same share of them, so the waves of an
//     XCD pass through the windows together (within a window or two) without waiting for each other: at any moment the
//     launch gathers from two or three windows of the source range, and a source piece is fetched past L2 once per XCD that
//     reads it (parts times in all) instead of once per neighbour. The windows are a matter of speed, not of correctness.
// A row is summed over its neighbours in CSR (ascending column) order with fmaf, one accumulator -- windows ascend and a
// window's entries of a row stay in column order: a fixed order, bitwise reproducible, the order of an unsplit row of the tile hop (rows the tile hop
// cuts into tiers are summed in another fixed order there).
// Placement: workgroup b takes the role of XCD b % 8 (workgroups b and b + 8 share an XCD on this chip; another placement
// costs speed only).
#include "common.h"
#include <cmath>

namespace elimrec {

struct SweepArgs {
    const int64_t *slot_ptr;      // [n_blocks * waves + 1]: a wave's stretch of the record stream, in STEPS (G records each)
    const uint4 *rec;             // the entry streams as 80-byte step records, [5 x uint4] each: 8 source pieces (source row x LPR: the
                                  // piece's float4 index in its slab) | 8 values | 8 rows relative to the block's first row as 16-bit
                                  // numbers; a slot without an entry reads source piece 0 with value 0 into row `dummy`
    int dummy;                    // = the launch's largest block: an LDS row behind every block's rows
    int64_t n_rows, n_src;        // rows of Xout / Xin
    const int32_t *block_ptr;     // [parts * passes * bpx + 1]: first row of every row block
    int parts, passes, bpx;       // row parts per slab (XCDs per slab), blocks a workgroup takes one after the other, workgroups per XCD
    int ns, nsx;                  // slabs; slabs taken side by side by the 8 XCD roles (min(ns, 8))
    const float4 *Xin;
    float4 *Xout;
    const float4 *Add;
    const uint32_t *add_mask;
    float scale;
    // ADAM: the swept rows' sums are the gradient of the table p_in, consumed here (the epilogue of the tile hop's Adam launch,
    // slab.hip stream_epilogue, element for element); Xout then is nullable (the gradient is written only when asked for)
    const float4 *ad_p_in;
    float4 *ad_p_out, *ad_m, *ad_v;
    float ad_step_size, ad_inv_sqrt_bc2, ad_beta1, ad_beta2, ad_eps, ad_wd;
};

template <int LPR, bool ADAM>
__global__ __launch_bounds__(64 * LPR) void sweep_rows_kernel(SweepArgs a) {
    constexpr int G = 64 / LPR;                         // lane groups of a wave
    constexpr int NT = 64 * LPR;                        // 64 lane groups per workgroup
    constexpr int NW = NT / 64;                         // waves
    extern __shared__ float4 acc[];                     // [rows of the block][LPR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane / LPR, cl = lane % LPR;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;  // XCD role, workgroup of the role
    const int part = x / a.nsx;
    for (int slab = x % a.nsx; slab < a.ns; slab += a.nsx) {
        const float4 *X = a.Xin + (int64_t)slab * a.n_src * LPR + cl;
        for (int pass = 0; pass < a.passes; ++pass) {
            const int blk = (part * a.passes + pass) * a.bpx + j;
            const int r0 = a.block_ptr[blk], nr = a.block_ptr[blk + 1] - r0;
            for (int r = tid; r < nr * LPR; r += NT) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tid < LPR) acc[a.dummy * LPR + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();
            // the wave's steps [e0, e1): step e is G records side by side, one per lane group (every lane of a group loads its group's
            // whole record); the records of the next PF - 1 steps are in flight while this one's pieces are gathered. No slot of a
            // record is predicated: empty ones gather source piece 0 with value 0 into the dummy row.
            const int64_t e0 = a.slot_ptr[(int64_t)blk * NW + wave], e1 = a.slot_ptr[(int64_t)blk * NW + wave + 1];
            constexpr int PF = 4;                          // records in flight (an even number: the gathers' two buffers alternate)
            uint4 rq[PF][5];
            auto load_rec = [&](uint4 (&dst)[5], int64_t e) {
                // (beyond the wave's stretch: the stream's closing records -- the plan appends PF all-empty steps)
                const uint4 *q = a.rec + (e * G + grp) * 5;
#pragma unroll
                for (int k = 0; k < 5; ++k) dst[k] = q[k];
            };
            auto gather = [&](const uint4 (&r)[5], float4 (&xs)[8]) {
                const uint32_t cs[8] = {r[0].x, r[0].y, r[0].z, r[0].w, r[1].x, r[1].y, r[1].z, r[1].w};
#pragma unroll
                for (int t = 0; t < 8; ++t) xs[t] = X[(int64_t)cs[t]];
            };
            auto step = [&](const uint4 (&r)[5], const float4 (&xs)[8]) {
                const uint32_t vb[8] = {r[2].x, r[2].y, r[2].z, r[2].w, r[3].x, r[3].y, r[3].z, r[3].w};
                const uint32_t rw[4] = {r[4].x, r[4].y, r[4].z, r[4].w};
                int rs[8];
                float4 as[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    rs[t] = (int)((t & 1) ? (rw[t >> 1] >> 16) : (rw[t >> 1] & 0xffffu));
                    as[t] = acc[rs[t] * LPR + cl];           // (every entry reads its row's sum; the later entries of a row take the running one)
                }
                float4 s = as[0];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (t > 0 && rs[t] != rs[t - 1]) s = as[t];
                    const float v = __uint_as_float(vb[t]);
                    s.x = fmaf(v, xs[t].x, s.x); s.y = fmaf(v, xs[t].y, s.y);
                    s.z = fmaf(v, xs[t].z, s.z); s.w = fmaf(v, xs[t].w, s.w);
                    acc[rs[t] * LPR + cl] = s;               // (a row's later entries overwrite with the longer sum: no test, no branch)
                }
            };
            // the pieces of step e + 1 are gathered while step e is added up: sixteen gathers in flight per lane
            float4 xa[8], xb[8];
#pragma unroll
            for (int d = 0; d < PF; ++d) load_rec(rq[d], e0 + d);
            gather(rq[0], xa);
            int64_t e = e0;
            for (; e + PF <= e1; e += PF) {
#pragma unroll
                for (int d = 0; d < PF; d += 2) {
                    gather(rq[d + 1], xb);
                    step(rq[d], xa);
                    load_rec(rq[d], e + PF + d);
                    gather(rq[(d + 2) % PF], xa);          // (d + 2 = PF: the record loaded a moment ago for step e + PF)
                    step(rq[d + 1], xb);
                    load_rec(rq[d + 1], e + PF + d + 1);
                }
            }
            // (the last, partial round: the records beyond e1 are the stream's empty closing steps)
#pragma unroll
            for (int d = 0; d < PF; d += 2) {
                if (e + d < e1) { gather(rq[d + 1], xb); step(rq[d], xa); }
                if (e + d + 1 < e1) { if (d + 2 < PF) gather(rq[d + 2], xa); step(rq[d + 1], xb); }
            }
            __syncthreads();
            for (int q = tid; q < nr * LPR; q += NT) {
                const int r = q / LPR, c4 = q % LPR;
                const int64_t row = r0 + r;
                const int64_t idx = ((int64_t)slab * a.n_rows + row) * LPR + c4;
                float4 v = acc[q];
                if (a.Add && (!a.add_mask || ((a.add_mask[row >> 5] >> (row & 31)) & 1u))) {
                    const float4 t = a.Add[idx];
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
                const float4 g = make_float4(v.x * a.scale, v.y * a.scale, v.z * a.scale, v.w * a.scale);
                if (ADAM) {
                    typedef float f4v __attribute__((ext_vector_type(4)));       // (each touched once per step: streamed past L2, as the tile form does)
                    const f4v p4 = __builtin_nontemporal_load((const f4v *)a.ad_p_in + idx), m4 = __builtin_nontemporal_load((const f4v *)a.ad_m + idx),
                              v4 = __builtin_nontemporal_load((const f4v *)a.ad_v + idx);
                    const float gr[4] = {g.x, g.y, g.z, g.w};
                    float po[4], mi[4], vi[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float gi = fmaf(a.ad_wd, p4[i], gr[i]);
                        mi[i] = m4[i] + (1.f - a.ad_beta1) * (gi - m4[i]);
                        vi[i] = fmaf(1.f - a.ad_beta2, gi * gi, a.ad_beta2 * v4[i]);
                        const float denom = sqrtf(vi[i]) * a.ad_inv_sqrt_bc2 + a.ad_eps;
                        po[i] = p4[i] - a.ad_step_size * (mi[i] / denom);
                    }
                    __builtin_nontemporal_store((f4v){mi[0], mi[1], mi[2], mi[3]}, (f4v *)a.ad_m + idx);
                    __builtin_nontemporal_store((f4v){vi[0], vi[1], vi[2], vi[3]}, (f4v *)a.ad_v + idx);
                    a.ad_p_out[idx] = make_float4(po[0], po[1], po[2], po[3]);
                    if (!a.Xout) continue;
                }
                a.Xout[idx] = g;
            }
            __syncthreads();
        }
    }
}

}  // namespace elimrec

using namespace elimrec;

extern "C" size_t elimrec_slab_sweep_lds_rows(int w) {
    // rows of a block: 160 KB of LDS per workgroup less a margin, a row piece of w floats each (+ the dummy row); row numbers
    // inside a block are 16-bit
    const size_t rows = (size_t)(156 * 1024) / ((size_t)w * 4) - 1;
    return rows < 65534 ? rows : 65534;
}

static int sweep_launch(const char *who, const int64_t *d_slot_ptr, const void *d_records, int64_t n_rows, int64_t n_src,
                        const int32_t *d_block_ptr, int parts, int passes, int bpx, int max_block_rows, int ns, int w, const float *d_Xin,
                        float *d_Xout, const float *d_add, const uint32_t *d_add_mask, float scale, const SweepArgs *adam, void *stream) {
    ELIMREC_REQUIRE(d_slot_ptr && d_records && d_block_ptr && d_Xin && (d_Xout || adam), "%s: null pointer", who);
    ELIMREC_REQUIRE(d_Xin != d_Xout, "%s: Xout must not alias Xin", who);
    ELIMREC_REQUIRE(w == 32 || w == 16, "%s: slab width %d (32 or 16 floats)", who, w);
    ELIMREC_REQUIRE(ns >= 1 && (ns <= 8 ? 8 % ns == 0 : ns % 8 == 0), "%s: %d slabs do not tile the 8 XCD roles", who, ns);
    const int nsx = ns < 8 ? ns : 8;
    ELIMREC_REQUIRE(parts == 8 / nsx && passes >= 1 && bpx >= 1, "%s: bad partition (parts %d passes %d bpx %d)", who, parts, passes, bpx);
    ELIMREC_REQUIRE(max_block_rows >= 1 && (size_t)max_block_rows <= elimrec_slab_sweep_lds_rows(w), "%s: a block of %d rows does not fit LDS",
                    who, max_block_rows);
    ELIMREC_REQUIRE(n_src * (w / 4) < ((int64_t)1 << 32), "%s: source piece numbers must fit 32 bits", who);
    SweepArgs a = adam ? *adam : SweepArgs{};
    a.slot_ptr = d_slot_ptr; a.rec = (const uint4 *)d_records; a.dummy = max_block_rows; a.n_rows = n_rows; a.n_src = n_src;
    a.block_ptr = d_block_ptr; a.parts = parts; a.passes = passes; a.bpx = bpx; a.ns = ns; a.nsx = nsx;
    a.Xin = (const float4 *)d_Xin; a.Xout = (float4 *)d_Xout; a.Add = (const float4 *)d_add; a.add_mask = d_add_mask; a.scale = scale;
    const size_t lds = (size_t)(max_block_rows + 1) * (size_t)w * 4;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(8 * bpx));
#define ELIMREC_SWEEP(LPR_, ADAM_)                                                                                                       \
    do {                                                                                                                                 \
        static bool attr = false;                                                                                                        \
        if (!attr) {                                                                                                                     \
            (void)hipFuncSetAttribute((const void *)sweep_rows_kernel<LPR_, ADAM_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr = true;                                                                                                                 \
        }                                                                                                                                \
        hipLaunchKernelGGL((sweep_rows_kernel<LPR_, ADAM_>), grid, dim3(64 * LPR_), lds, s, a);                                          \
    } while (0)
    if (w == 32) { if (adam) ELIMREC_SWEEP(8, true); else ELIMREC_SWEEP(8, false); }
    else { if (adam) ELIMREC_SWEEP(4, true); else ELIMREC_SWEEP(4, false); }
#undef ELIMREC_SWEEP
    return check_hip(hipGetLastError(), who);
}

extern "C" int elimrec_slab_sweep_hop(const int64_t *d_slot_ptr, const void *d_records,
                                      int64_t n_rows, int64_t n_src, const int32_t *d_block_ptr, int parts, int passes, int bpx,
                                      int max_block_rows, int ns, int w, const float *d_Xin, float *d_Xout,
                                      const float *d_add, const uint32_t *d_add_mask, float scale, void *stream) {
    return sweep_launch("slab_sweep_hop", d_slot_ptr, d_records, n_rows, n_src, d_block_ptr, parts, passes, bpx, max_block_rows, ns, w,
                        d_Xin, d_Xout, d_add, d_add_mask, scale, nullptr, stream);
}

// The same sweep as the adjoint's LAST hop: the swept rows' sums are the gradient of the fp32 table d_p_in, consumed by the Adam
// step in the launch's epilogue (the arithmetic of elimrec_slab_hop_adam / elimrec_adam_multi, element for element; coupled L2,
// 1-based step); d_grad_out nullable.
extern "C" int elimrec_slab_sweep_hop_adam(const int64_t *d_slot_ptr, const void *d_records, int64_t n_rows, int64_t n_src,
                                           const int32_t *d_block_ptr, int parts, int passes, int bpx, int max_block_rows, int ns, int w,
                                           const float *d_Xin, float *d_grad_out, const float *d_add, const uint32_t *d_add_mask,
                                           float scale, const float *d_p_in, float *d_p_out, float *d_m, float *d_v, float lr,
                                           float beta1, float beta2, float eps, float weight_decay, int64_t step, void *stream) {
    ELIMREC_REQUIRE(d_p_in && d_p_out && d_m && d_v, "slab_sweep_hop_adam: null pointer");
    ELIMREC_REQUIRE(step >= 1, "slab_sweep_hop_adam: 1-based step");
    ELIMREC_REQUIRE((const void *)d_Xin != (const void *)d_p_out && (const void *)d_Xin != (const void *)d_m &&
                        (const void *)d_Xin != (const void *)d_v, "slab_sweep_hop_adam: the gathered table must not be written");
    SweepArgs ad = {};
    ad.ad_p_in = (const float4 *)d_p_in; ad.ad_p_out = (float4 *)d_p_out; ad.ad_m = (float4 *)d_m; ad.ad_v = (float4 *)d_v;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    ad.ad_step_size = (float)((double)lr / bc1);
    ad.ad_inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    ad.ad_beta1 = beta1; ad.ad_beta2 = beta2; ad.ad_eps = eps; ad.ad_wd = weight_decay;
    return sweep_launch("slab_sweep_hop_adam", d_slot_ptr, d_records, n_rows, n_src, d_block_ptr, parts, passes, bpx, max_block_rows, ns, w,
                        d_Xin, d_grad_out, d_add, d_add_mask, scale, &ad, stream);
}
