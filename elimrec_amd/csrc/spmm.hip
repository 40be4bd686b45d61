// LightGCN propagation for all M tables at once (models/EliMRec.py:238-248, K2-K4 of SURVEY §2a).
//
// Layout: one [N x C] fp32 table, C = M*d; at the Tiktok shape a row is 1 KiB = one wave x
// float4, so a neighbour row is fetched by ONE fully coalesced wave instruction and one pass
// over the CSR structure serves every table. HBM-bound: per hop read C*4 B per non-zero
// (+ 8 B of CSR), write C*4 B per row.
#include "common.h"
#include <cstdlib>

namespace elimrec {

// X0[u, m*d + j] = user_emb[u, j]; X0[U+i, j] = item_emb[i, j] (j < d)
__global__ void assemble_x0_kernel(const float4 *__restrict__ ue, const float4 *__restrict__ ie,
                                   float4 *__restrict__ X0, int64_t U, int64_t I, int d4, int M) {
    const int64_t row = blockIdx.x;
    const int C4 = d4 * M;
    if (row < U) {
        for (int c = threadIdx.x; c < C4; c += blockDim.x) X0[row * C4 + c] = ue[row * d4 + (c % d4)];
    } else {
        const int64_t i = row - U;
        for (int c = threadIdx.x; c < d4; c += blockDim.x) X0[row * C4 + c] = ie[i * d4 + c];
    }
}

// One wave per output row; lane l owns float4 column l of each 256-column chunk.
// Neighbour (col, val) pairs are wave-uniform; UNROLL rows are kept in flight per lane.
template <int UNROLL>
__global__ __launch_bounds__(256) void spmm_hop_kernel(const int32_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ col,
                                                       const float *__restrict__ val, int64_t n_rows, int C4,
                                                       const float4 *__restrict__ Xin, float4 *__restrict__ Xout,
                                                       const float4 *__restrict__ AccIn, float4 *__restrict__ AccOut,
                                                       float scale, int long_threshold) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    if (end - beg > long_threshold) return;     // split rows: spmm_long_partial + spmm_long_fixup
    for (int c0 = 0; c0 < C4; c0 += 64) {
        const int c = c0 + lane;
        const bool on = c < C4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // UNROLL neighbour rows in flight; the tail is handled by wave-uniform guards (j, end are
        // uniform) so a short row still issues all its loads back to back.
        for (int j = beg; j < end; j += UNROLL) {
            int cj[UNROLL];
            float vj[UNROLL];
            float4 x[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const bool in = (j + u) < end;
                cj[u] = in ? col[j + u] : 0;
                vj[u] = in ? val[j + u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                x[u] = (on && (j + u) < end) ? Xin[(int64_t)cj[u] * C4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                acc.x = fmaf(vj[u], x[u].x, acc.x); acc.y = fmaf(vj[u], x[u].y, acc.y);
                acc.z = fmaf(vj[u], x[u].z, acc.z); acc.w = fmaf(vj[u], x[u].w, acc.w);
            }
        }
        if (on) {
            if (Xout) Xout[row * C4 + c] = acc;
            if (AccOut) {
                const float4 a = AccIn[row * C4 + c];
                AccOut[row * C4 + c] = make_float4((a.x + acc.x) * scale, (a.y + acc.y) * scale,
                                                   (a.z + acc.z) * scale, (a.w + acc.w) * scale);
            }
        }
    }
}

// Rows with more than `long_threshold` non-zeros (the power-law head) are cut into segments of at
// most that many; one wave sums one segment into partials[seg], then one wave per long row adds
// its partials in segment order and applies the epilogue. Fixed order => bitwise reproducible.
template <int UNROLL>
__global__ __launch_bounds__(256) void spmm_long_partial_kernel(const int32_t *__restrict__ seg_bounds, int n_seg,
                                                                const int32_t *__restrict__ col,
                                                                const float *__restrict__ val, int C4,
                                                                const float4 *__restrict__ Xin,
                                                                float4 *__restrict__ partials) {
    const int lane = threadIdx.x & 63;
    const int seg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (seg >= n_seg) return;
    const int beg = seg_bounds[2 * seg], end = seg_bounds[2 * seg + 1];
    for (int c0 = 0; c0 < C4; c0 += 64) {
        const int c = c0 + lane;
        const bool on = c < C4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // UNROLL neighbour rows in flight; the tail is handled by wave-uniform guards (j, end are
        // uniform) so a short row still issues all its loads back to back.
        for (int j = beg; j < end; j += UNROLL) {
            int cj[UNROLL];
            float vj[UNROLL];
            float4 x[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const bool in = (j + u) < end;
                cj[u] = in ? col[j + u] : 0;
                vj[u] = in ? val[j + u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                x[u] = (on && (j + u) < end) ? Xin[(int64_t)cj[u] * C4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                acc.x = fmaf(vj[u], x[u].x, acc.x); acc.y = fmaf(vj[u], x[u].y, acc.y);
                acc.z = fmaf(vj[u], x[u].z, acc.z); acc.w = fmaf(vj[u], x[u].w, acc.w);
            }
        }
        if (on) partials[(int64_t)seg * C4 + c] = acc;
    }
}

__global__ __launch_bounds__(256) void spmm_long_fixup_kernel(const int32_t *__restrict__ long_rows, int n_long,
                                                              const int32_t *__restrict__ long_seg_ptr, int C4,
                                                              const float4 *__restrict__ partials,
                                                              float4 *__restrict__ Xout,
                                                              const float4 *__restrict__ AccIn,
                                                              float4 *__restrict__ AccOut, float scale) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n_long) return;
    const int64_t row = long_rows[i];
    const int sb = long_seg_ptr[i], se = long_seg_ptr[i + 1];
    for (int c = lane; c < C4; c += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int sgm = sb;
        for (; sgm + 8 <= se; sgm += 8) {          // 8 partial rows in flight, summed in segment order
            float4 p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = partials[(int64_t)(sgm + u) * C4 + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += p[u].x; acc.y += p[u].y; acc.z += p[u].z; acc.w += p[u].w; }
        }
        for (; sgm < se; ++sgm) {
            const float4 p = partials[(int64_t)sgm * C4 + c];
            acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
        if (Xout) Xout[row * C4 + c] = acc;
        if (AccOut) {
            const float4 a = AccIn[row * C4 + c];
            AccOut[row * C4 + c] = make_float4((a.x + acc.x) * scale, (a.y + acc.y) * scale,
                                               (a.z + acc.z) * scale, (a.w + acc.w) * scale);
        }
    }
}

__global__ void scale_copy_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, int64_t n4, float s) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = in[i];
        out[i] = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
    }
}

}  // namespace elimrec

using namespace elimrec;

extern "C" int elimrec_assemble_x0(const float *d_user_emb, const float *d_item_emb, float *d_X0, int64_t U,
                                   int64_t I, int d, int M, void *stream) {
    ELIMREC_REQUIRE(d_user_emb && d_item_emb && d_X0, "assemble_x0: null pointer");
    ELIMREC_REQUIRE(d > 0 && d % 4 == 0 && M >= 1, "assemble_x0: recdim must be a positive multiple of 4");
    if (U + I == 0) return 0;
    hipLaunchKernelGGL(assemble_x0_kernel, dim3((unsigned)(U + I)), dim3(64), 0, (hipStream_t)stream,
                       (const float4 *)d_user_emb, (const float4 *)d_item_emb, (float4 *)d_X0, U, I, d / 4, M);
    ELIMREC_LAUNCH_CHECK("assemble_x0");
    return 0;
}

static int spmm_hop_impl(const int32_t *d_rowptr, const int32_t *d_col, const float *d_val, int64_t n_rows, int C,
                         const elimrec_csr_split *split, const float *d_Xin, float *d_Xout, const float *d_AccIn,
                         float *d_AccOut, float scale, void *stream) {
    ELIMREC_REQUIRE(d_rowptr && d_Xin, "spmm_hop: null pointer");
    ELIMREC_REQUIRE(C > 0 && C % 4 == 0, "spmm_hop: C must be a positive multiple of 4");
    ELIMREC_REQUIRE(d_Xout || d_AccOut, "spmm_hop: no output");
    ELIMREC_REQUIRE(!d_AccOut || d_AccIn, "spmm_hop: AccOut needs AccIn");
    ELIMREC_REQUIRE(d_Xout != d_Xin, "spmm_hop: Xout must not alias Xin");
    if (n_rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool has_split = split && split->n_long > 0;
    if (has_split)
        ELIMREC_REQUIRE(split->d_long_rows && split->d_long_seg_ptr && split->d_seg_bounds && split->d_partials &&
                            split->n_seg > 0 && split->long_threshold > 0,
                        "spmm_hop: incomplete row-split plan");
    const int thr = has_split ? split->long_threshold : INT32_MAX;
    const int waves = 4;
    hipLaunchKernelGGL(spmm_hop_kernel<8>, dim3((unsigned)((n_rows + waves - 1) / waves)), dim3(64 * waves), 0, s,
                       d_rowptr, d_col, d_val, n_rows, C / 4, (const float4 *)d_Xin, (float4 *)d_Xout,
                       (const float4 *)d_AccIn, (float4 *)d_AccOut, scale, thr);
    ELIMREC_LAUNCH_CHECK("spmm_hop");
    if (has_split) {
        hipLaunchKernelGGL(spmm_long_partial_kernel<8>, dim3((unsigned)((split->n_seg + waves - 1) / waves)),
                           dim3(64 * waves), 0, s, split->d_seg_bounds, split->n_seg, d_col, d_val, C / 4,
                           (const float4 *)d_Xin, (float4 *)split->d_partials);
        ELIMREC_LAUNCH_CHECK("spmm_long_partial");
        hipLaunchKernelGGL(spmm_long_fixup_kernel, dim3((unsigned)((split->n_long + waves - 1) / waves)),
                           dim3(64 * waves), 0, s, split->d_long_rows, split->n_long, split->d_long_seg_ptr, C / 4,
                           (const float4 *)split->d_partials, (float4 *)d_Xout, (const float4 *)d_AccIn,
                           (float4 *)d_AccOut, scale);
        ELIMREC_LAUNCH_CHECK("spmm_long_fixup");
    }
    return 0;
}

extern "C" int elimrec_spmm_hop(const int32_t *d_rowptr, const int32_t *d_col, const float *d_val, int64_t n_rows,
                                int C, const elimrec_csr_split *split, const float *d_Xin, float *d_Xout,
                                const float *d_AccIn, float *d_AccOut, float scale, void *stream) {
    return spmm_hop_impl(d_rowptr, d_col, d_val, n_rows, C, split, d_Xin, d_Xout, d_AccIn, d_AccOut, scale, stream);
}

extern "C" int elimrec_propagate(const int32_t *d_rowptr, const int32_t *d_col, const float *d_val, int64_t n_rows,
                                 int C, const elimrec_csr_split *split, int L, const float *d_X0, float *d_tmp0,
                                 float *d_tmp1, float *d_Out, void *stream) {
    ELIMREC_REQUIRE(L >= 0, "propagate: layer_num must be >= 0");
    ELIMREC_REQUIRE(d_X0 && d_Out && d_Out != d_X0, "propagate: bad X0/Out");
    ELIMREC_REQUIRE(C > 0 && C % 4 == 0, "propagate: C must be a positive multiple of 4");
    const float inv = 1.0f / (float)(L + 1);
    if (L == 0) {
        const int64_t n4 = n_rows * (C / 4);
        if (n4 == 0) return 0;
        hipLaunchKernelGGL(scale_copy_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const float4 *)d_X0,
                           (float4 *)d_Out, n4, 1.0f);
        ELIMREC_LAUNCH_CHECK("scale_copy");
        return 0;
    }
    ELIMREC_REQUIRE(L < 2 || d_tmp0, "propagate: tmp0 required for L >= 2");
    ELIMREC_REQUIRE(L < 3 || d_tmp1, "propagate: tmp1 required for L >= 3");
    // hop k reads X^{k-1}, writes X^k (except the last hop) and folds X^k into the running sum
    // kept in Out; the last hop applies 1/(L+1).
    const float *xin = d_X0;
    float *bufs[2] = {d_tmp0, d_tmp1};
    for (int k = 1; k <= L; ++k) {
        const bool last = (k == L);
        float *xout = last ? nullptr : bufs[(k - 1) & 1];
        const float *accin = (k == 1) ? d_X0 : d_Out;
        int rc = spmm_hop_impl(d_rowptr, d_col, d_val, n_rows, C, split, xin, xout, accin, d_Out, last ? inv : 1.0f,
                               stream);
        if (rc) return rc;
        xin = xout;
    }
    return 0;
}

// =====================================================================================================
// Bipartite propagation.
//
// The 'pre' / 'plain' / 'gcmc' adjacencies have no diagonal: A = [[0, P], [Q, 0]] (P: users <- items,
// Q: items <- users). Layer k of table m is then alive on ONE side only once the layer-0 table is split
// as [E_u ; item_m] = [E_u ; 0] + [0 ; item_m]:
//   * the "wide" chain  w_k = A^k [0 ; XI]   (all M tables, C = M*d columns) lives on users for odd k,
//     on items for even k;
//   * the "narrow" chain a_k = A^k [E_u ; 0] is the SAME for every table (d columns) and lives on items
//     for odd k, users for even k.
// One reference hop (every row gathers C columns over all nnz) becomes two half hops: nnz/2 edges at C
// columns + nnz/2 edges at d columns = 62.5 % of the gather traffic for M = 4, with identical results
// (Out = 1/(L+1) sum_k (w_k + bcast(a_k)) on every row). The backward pass is the adjoint in Horner form:
//   gXI  = inv (G_i + P^T (G_u + Q^T (G_i + P^T G_u)))          L half hops at C columns
//   gE_u = inv (H_u + Q^T (H_i + P^T (H_u + Q^T H_i)))          L half hops at d columns, H = blocksum(G)
// and the first half hop of each gathers from the raw head gradient, which is non-zero on <= 3B rows:
// a row bitmap skips the rest (and the zero-fill of G disappears with it).
// =====================================================================================================
extern "C" int elimrec_ticket_fixup(void);

namespace elimrec {

struct HalfArgs {
    const int32_t *rowptr, *col;
    const float *val;
    int64_t n_rows;
    int W4;                       // columns processed, in float4
    int ld4;                      // row stride of Xin / Xout / Add2 in float4 (>= W4)
    int ld_add1;                  // row stride of Add1 in float4
    int ld_acc;                   // row stride of AccOut in float4
    // second accumulator, rows [acc2_lo, acc2_hi) only (contiguous, stride W4):
    //   Acc2Out[row] = (r + Acc2In[row]) * acc2_scale         (Acc2In nullable)
    float4 *Acc2Out;
    const float4 *Acc2In;
    int64_t acc2_lo, acc2_hi;
    float acc2_scale;
    const float4 *Xin;
    const uint32_t *src_mask;     // nullable: bit r set <=> source row r is non-zero
    float4 *Xout;                 // nullable: raw result
    const float4 *Add1;           // nullable addends of the epilogue (own row)
    const uint32_t *add1_mask;    // nullable: Add1 row is read only if its bit is set
    const float4 *Add2;
    const float4 *AddN;           // nullable narrow addend [n_rows x N4], broadcast over the W4/N4 blocks
    int N4;
    float4 *AccOut;               // nullable: (r + Add1 + Add2 + bcast(AddN)) * scale
    float scale;
    int long_threshold;
    // long-row plan
    const int32_t *seg_bounds;
    int n_seg;
    const int32_t *long_rows;
    int n_long;
    const int32_t *long_seg_ptr;
    float4 *partials;
    const int32_t *row_order;     // nullable [n_rows]: processing order of the rows (degree-sorted so the rows that
                                  // share a wave have similar lengths); results do not depend on it
    const int32_t *seg_row;       // [n_seg] index (into long_rows) of the split row a segment belongs to
    int32_t *tickets;             // [n_long] arrival counters, zero between launches (self-resetting)
    const int32_t *row_items;     // nullable [n_row_items][3]: (row, begin, end) of the unsplit rows in processing order
    int n_row_items;
    // nullable: only the rows of this device-side list are processed (first min(*row_list_count, row_list_cap)
    // entries; split rows are still all done by the segment workgroups). The other rows of the outputs are not written.
    const int32_t *row_list;
    const int32_t *row_list_count;
    int64_t row_list_cap;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool mask_bit(const uint32_t *m, int r) { return (m[r >> 5] >> (r & 31)) & 1u; }

__device__ __forceinline__ void half_epilogue(const HalfArgs &a, int64_t row, int c, float4 r) {
    if (a.Xout) a.Xout[row * a.ld4 + c] = r;
    if (a.AccOut) {
        float4 s = r;
        if (a.Add1 && (!a.add1_mask || mask_bit(a.add1_mask, (int)row))) {
            const float4 t = a.Add1[row * a.ld_add1 + c];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        if (a.Add2) {
            const float4 t = a.Add2[row * a.ld4 + c];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        if (a.AddN) {
            const float4 t = a.AddN[row * a.N4 + (c % a.N4)];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        a.AccOut[row * a.ld_acc + c] = make_float4(s.x * a.scale, s.y * a.scale, s.z * a.scale, s.w * a.scale);
    }
    if (a.Acc2Out && row >= a.acc2_lo && row < a.acc2_hi) {
        float4 s = r;
        if (a.Acc2In) {
            const float4 t = a.Acc2In[row * a.W4 + c];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        a.Acc2Out[row * a.W4 + c] = make_float4(s.x * a.acc2_scale, s.y * a.acc2_scale, s.z * a.acc2_scale, s.w * a.acc2_scale);
    }
}

// sum_j val[j] * Xin[col[j], c] over [beg, end), UNROLL rows in flight, optional source-row bitmap.
template <int UNROLL>
__device__ __forceinline__ float4 half_gather(const HalfArgs &a, int beg, int end, int c, bool on) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = beg; j < end; j += UNROLL) {
        int cj[UNROLL];
        float vj[UNROLL];
        bool in[UNROLL];
        float4 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            in[u] = (j + u) < end;
            cj[u] = in[u] ? a.col[j + u] : 0;
            vj[u] = in[u] ? a.val[j + u] : 0.f;
        }
        if (a.src_mask) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) in[u] = in[u] && mask_bit(a.src_mask, cj[u]);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            x[u] = (on && in[u]) ? a.Xin[(int64_t)cj[u] * a.ld4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            acc.x = fmaf(vj[u], x[u].x, acc.x); acc.y = fmaf(vj[u], x[u].y, acc.y);
            acc.z = fmaf(vj[u], x[u].z, acc.z); acc.w = fmaf(vj[u], x[u].w, acc.w);
        }
    }
    return acc;
}

// Same sum when the source table is row-sparse (the raw head gradient: non-zero on <= 3B rows): the LPR
// lanes of a row test LPR neighbours at once against the row bitmap (one coalesced col load + one bitmap
// probe per lane + a ballot) and only the few active ones are gathered. Neighbour order is preserved, so
// the result equals half_gather's on the same data with zeros in the inactive rows.
template <int LPR>
__device__ __forceinline__ float4 half_gather_masked(const HalfArgs &a, int beg, int end, int c, bool on, int sub,
                                                     int cl) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j0 = beg; j0 < end; j0 += LPR) {
        const int jj = j0 + cl;
        const bool in = jj < end;
        const int cj = in ? a.col[jj] : 0;
        const float vj = in ? a.val[jj] : 0.f;
        const bool act = in && mask_bit(a.src_mask, cj);
        unsigned long long bits = __ballot(act);
        if (LPR < 64) bits = (bits >> (sub * LPR)) & ((1ull << LPR) - 1ull);
        while (bits) {
            const int b = __builtin_ctzll(bits);
            bits &= bits - 1ull;
            const int src = sub * LPR + b;
            const int cjb = __shfl(cj, src, 64);
            const float vjb = __shfl(vj, src, 64);
            if (on) {
                const float4 x = a.Xin[(int64_t)cjb * a.ld4 + c];
                acc.x = fmaf(vjb, x.x, acc.x); acc.y = fmaf(vjb, x.y, acc.y);
                acc.z = fmaf(vjb, x.z, acc.z); acc.w = fmaf(vjb, x.w, acc.w);
            }
        }
    }
    return acc;
}

// Sum of one split row's partial rows by a WHOLE wave: the wave's 64/LPR lane groups each add a contiguous
// quarter of the segments in order (8 loads in flight), the quarter sums are added in group order through
// shuffles, group 0 runs the epilogue. A fixed order, so the in-launch combine and half_fixup_kernel give the
// same bits; the longest row (154 segments at the Tiktok shape) needs 5 dependent load rounds instead of 20.
// Must be called by all lanes of the wave with wave-uniform li.
template <int LPR>
__device__ __forceinline__ void combine_split_row(const HalfArgs &a, int li) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int q = lane / LPR, cl = lane % LPR;
    const int64_t row = a.long_rows[li];
    const int sb = a.long_seg_ptr[li], se = a.long_seg_ptr[li + 1];
    const int per = (se - sb + RPW - 1) / RPW;
    const int qb = min(sb + q * per, se), qe = min(qb + per, se);
    for (int c0 = 0; c0 < a.W4; c0 += LPR) {
        const int c = c0 + cl;
        const bool con = c < a.W4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (con) {
            int sgm = qb;
            for (; sgm + 8 <= qe; sgm += 8) {
                float4 p[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = a.partials[(int64_t)(sgm + u) * a.W4 + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc.x += p[u].x; acc.y += p[u].y; acc.z += p[u].z; acc.w += p[u].w; }
            }
            for (; sgm < qe; ++sgm) {
                const float4 p = a.partials[(int64_t)sgm * a.W4 + c];
                acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
            }
        }
        float4 tot = acc;
        if (RPW > 1) {
            tot.x = __shfl(acc.x, cl, 64); tot.y = __shfl(acc.y, cl, 64); tot.z = __shfl(acc.z, cl, 64); tot.w = __shfl(acc.w, cl, 64);
#pragma unroll
            for (int t = 1; t < RPW; ++t) {
                tot.x += __shfl(acc.x, t * LPR + cl, 64); tot.y += __shfl(acc.y, t * LPR + cl, 64);
                tot.z += __shfl(acc.z, t * LPR + cl, 64); tot.w += __shfl(acc.w, t * LPR + cl, 64);
            }
        }
        if (con && q == 0) half_epilogue(a, row, c, tot);
    }
}

// The unsplit rows as a persistent stream (W4 <= LPR: one float4 column per lane). A row costs four dependent
// memory round trips when taken cold (order -> rowptr -> col/val -> source rows) and the rows are short (a dozen
// neighbours), so the kernel is bound by that chain times the rows a CU can keep in flight -- measured: 33 us per
// hop with the whole source table in L2 against 45 us from HBM. Here every wave walks its items k, k + stride, ...
// with the index chain software-pipelined: the (row, begin, end) triple of item k+2 and the first UNROLL
// (col, val) of item k+1 are in flight while item k gathers. Sums are formed in the same order with the same
// fmaf as half_gather, so the results are bit-identical.
template <int LPR, int UNROLL>
__device__ __forceinline__ void half_stream_rows(const HalfArgs &a, int seg_blocks) {
    constexpr int RPW = 64 / LPR;
    static_assert(LPR >= UNROLL && LPR % UNROLL == 0, "one (col, val) per lane: a lane group holds LPR neighbours");
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, cl = lane % LPR;
    const int wpb = blockDim.x >> 6;
    const int64_t wave_g = (int64_t)((int)blockIdx.x - seg_blocks) * wpb + (threadIdx.x >> 6);
    const int64_t n_slots = (int64_t)((int)gridDim.x - seg_blocks) * wpb;
    const int64_t n_items = a.n_row_items;
    const bool con = cl < a.W4;
    int crow, cbeg, cend, nrow, nbeg, nend;
    auto load_trip = [&](int64_t k, int &row, int &beg, int &end) {
        const int64_t it = (wave_g + k * n_slots) * RPW + sub;
        row = -1; beg = 0; end = 0;
        if (it < n_items) {
            const int32_t *p = a.row_items + 3 * it;
            row = p[0]; beg = p[1]; end = p[2];
        }
    };
    // neighbour indices: lane cl of a group holds neighbour (base + cl) of the group's row -- ONE coalesced load per
    // LPR neighbours instead of UNROLL broadcast loads per UNROLL neighbours; the gather reads them by shuffle
    int mc, nmc;
    float mv, nmv;
    auto load_idx = [&](int j, int end, int &c, float &v) {
        const bool in = j + cl < end;
        c = in ? a.col[j + cl] : 0;
        v = in ? a.val[j + cl] : 0.f;
    };
    load_trip(0, crow, cbeg, cend);
    load_trip(1, nrow, nbeg, nend);
    load_idx(cbeg, cend, mc, mv);
    for (int64_t k = 0; (wave_g + k * n_slots) * RPW < n_items; ++k) {
        int frow, fbeg, fend;
        load_trip(k + 2, frow, fbeg, fend);
        load_idx(nbeg, nend, nmc, nmv);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // every lane walks the chunks of the LONGEST row of the wave (shuffles need all lanes); shorter rows add zeros
        int len = cend - cbeg;
#pragma unroll
        for (int o = 32; o >= LPR; o >>= 1) len = max(len, __shfl_xor(len, o, 64));
        for (int base = 0; base < len; base += UNROLL) {
            if (base > 0 && base % LPR == 0) load_idx(cbeg + base, cend, mc, mv);
            float4 x[UNROLL];
            float vj[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int src = sub * LPR + ((base + u) % LPR);
                const int cj = __shfl(mc, src, 64);
                vj[u] = __shfl(mv, src, 64);
                x[u] = (con && cbeg + base + u < cend) ? a.Xin[(int64_t)cj * a.ld4 + cl] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                acc.x = fmaf(vj[u], x[u].x, acc.x); acc.y = fmaf(vj[u], x[u].y, acc.y);
                acc.z = fmaf(vj[u], x[u].z, acc.z); acc.w = fmaf(vj[u], x[u].w, acc.w);
            }
        }
        if (con && crow >= 0) half_epilogue(a, crow, cl, acc);
        crow = nrow; cbeg = nbeg; cend = nend;
        nrow = frow; nbeg = fbeg; nend = fend;
        mc = nmc; mv = nmv;
    }
}

// LPR lanes per row (power of two <= 64): a wave handles 64/LPR rows, so narrow (d-column) tables use
// every lane. ONE launch covers both kinds of work item: the first `seg_blocks` workgroups take segments
// of the split (long) rows and write partial sums, the rest take whole rows of the CSR (long rows
// skipped) and finish them. The heavy segment waves start first and the two kinds overlap.
template <int LPR, int UNROLL, bool MASKED>
__global__ __launch_bounds__(256) void half_hop_kernel(HalfArgs a, int seg_blocks) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, cl = lane % LPR;
    const bool seg_mode = (int)blockIdx.x < seg_blocks;            // workgroup-uniform
    if constexpr (LPR >= UNROLL) {
        if (!MASKED && !seg_mode && a.row_items && !a.row_list && a.W4 <= LPR) {      // workgroup-uniform
            half_stream_rows<LPR, UNROLL>(a, seg_blocks);
            return;
        }
    }
    const int64_t blk = seg_mode ? blockIdx.x : (blockIdx.x - seg_blocks);
    int64_t item = (blk * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RPW + sub;
    const int64_t n_items = seg_mode ? (int64_t)a.n_seg
                                     : (a.row_list ? min((int64_t)*a.row_list_count, a.row_list_cap) : a.n_rows);
    bool valid = item < n_items;
    int beg = 0, end = 0;
    if (valid) {
        if (!seg_mode) {
            if (a.row_list) item = a.row_list[item];
            else if (a.row_order) item = a.row_order[item];
            beg = a.rowptr[item]; end = a.rowptr[item + 1];
            if (end - beg > a.long_threshold) valid = false;
        } else {
            beg = a.seg_bounds[2 * item]; end = a.seg_bounds[2 * item + 1];
        }
    }
    if (LPR == 64) {
        beg = __builtin_amdgcn_readfirstlane(beg);
        end = __builtin_amdgcn_readfirstlane(end);
    }
    if (!valid) { beg = 0; end = 0; }
    // partial rows are published write-through (sc1) when another wave will combine them in this launch
    const bool publish = seg_mode && a.tickets != nullptr;         // workgroup-uniform
    __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)a.partials, 0, publish ? (int)((size_t)a.n_seg * a.W4 * 16) : 0, 0x00020000);
    for (int c0 = 0; c0 < a.W4; c0 += LPR) {
        const int c = c0 + cl;
        const bool on = valid && c < a.W4;
        const float4 acc = MASKED ? half_gather_masked<LPR>(a, beg, end, c, on, sub, cl)
                                  : half_gather<UNROLL>(a, beg, end, c, on);
        if (on) {
            if (!seg_mode) half_epilogue(a, item, c, acc);
            else if (!publish) a.partials[item * a.W4 + c] = acc;
            else {
                u32x4 bits = {__float_as_uint(acc.x), __float_as_uint(acc.y), __float_as_uint(acc.z), __float_as_uint(acc.w)};
                __builtin_amdgcn_raw_buffer_store_b128(bits, prsrc, (unsigned)((item * a.W4 + c) * 16), 0, 16 /* sc1 */);
            }
        }
    }
    if (!publish) return;                                          // workgroup-uniform
    // ---- last arriver combines: every segment wave has stored its partial row write-through (sc1: no
    // release fence, which would write back the whole XCD L2 under the main rows' output stream), drains
    // its stores, draws a ticket for its split row, and the wave that draws the last one sums that row's
    // partials in the fixed order of combine_split_row (the one half_fixup_kernel uses => same bits) and runs the epilogue.
    // cdna_hip_programming.md Guideline 16, recipe R1 in its counter form: sc1 stores -> every storing wave
    // s_waitcnt vmcnt(0) -> relaxed agent-scope atomic add; consumer: ONE agent acquire (drops this CU's
    // stale L1 lines) -> s_waitcnt vmcnt(0) -> plain loads.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int li = -1, nseg = 0, ticket = -1;
    if (valid) {
        li = a.seg_row[item];
        nseg = a.long_seg_ptr[li + 1] - a.long_seg_ptr[li];
        if (cl == 0) ticket = __hip_atomic_fetch_add(&a.tickets[li], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ticket = __shfl(ticket, sub * LPR, 64);
    const bool last = valid && (ticket == nseg - 1);
    if (__ballot(last) == 0ull) return;                             // wave-uniform
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int g = 0; g < RPW; ++g) {                                 // one finished row at a time, the whole wave on it
        const int g_last = __shfl(last ? 1 : 0, g * LPR, 64);
        if (!g_last) continue;                                      // wave-uniform
        const int g_li = __shfl(li, g * LPR, 64);
        combine_split_row<LPR>(a, g_li);
        if (lane == 0) __hip_atomic_store(&a.tickets[g_li], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void half_fixup_kernel(HalfArgs a) {
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // one wave per split row
    if (i >= a.n_long) return;
    combine_split_row<LPR>(a, i);
}

// Out[row] = (Add1[row] + bcast(AddN[row])) * scale   (a side that has no wide step of its own: L <= 1)
__global__ void combine_kernel(const float4 *__restrict__ Add1, const float4 *__restrict__ AddN, int64_t n_rows,
                               int W4, int N4, float scale, float4 *__restrict__ out) {
    const int64_t total = n_rows * W4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / W4;
        const int c = (int)(t - row * W4);
        float4 s = Add1 ? Add1[t] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (AddN) {
            const float4 n = AddN[row * N4 + (c % N4)];
            s.x += n.x; s.y += n.y; s.z += n.z; s.w += n.w;
        }
        out[t] = make_float4(s.x * scale, s.y * scale, s.z * scale, s.w * scale);
    }
}

// bitmaps of the active (non-zero) rows on each side + H = blocksum of the active rows of G
__global__ void build_masks_kernel(const int32_t *__restrict__ active_rows, const int32_t *__restrict__ seg_info,
                                   int64_t n_max, int64_t U, uint32_t *__restrict__ mask_u,
                                   uint32_t *__restrict__ mask_i) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_max || s >= seg_info[0]) return;
    const int r = active_rows[s];
    if (r < U) atomicOr(&mask_u[r >> 5], 1u << (r & 31));
    else atomicOr(&mask_i[(r - U) >> 5], 1u << ((r - U) & 31));
}

static int stream_wgs() {
    static int v = 0;
    if (!v) {
        v = 3072;                          // 256 CUs x 4 resident workgroups x 3 (measured flat from 1536 up)
    }
    return v;
}

static int launch_half(const elimrec_csr *m, HalfArgs a, size_t partials_offset, hipStream_t s) {
    a.rowptr = m->d_rowptr; a.col = m->d_col; a.val = m->d_val; a.n_rows = m->n_rows;
    const bool has_split = m->split.n_long > 0;
    a.long_threshold = has_split ? m->split.long_threshold : INT32_MAX;
    a.seg_bounds = m->split.d_seg_bounds; a.n_seg = has_split ? m->split.n_seg : 0;
    a.long_rows = m->split.d_long_rows; a.n_long = has_split ? m->split.n_long : 0;
    a.long_seg_ptr = m->split.d_long_seg_ptr;
    a.partials = m->split.d_partials ? (float4 *)(m->split.d_partials + partials_offset) : nullptr;
    a.seg_row = m->split.d_seg_row;
    a.row_order = m->split.d_row_order;
    a.row_items = m->split.d_row_items;
    a.n_row_items = m->split.n_row_items;
    // the two chains may touch the same block concurrently: wide launches use tickets[0..n_long), narrow ones the next n_long
    a.tickets = (m->split.d_tickets && elimrec_ticket_fixup())
                    ? m->split.d_tickets + (partials_offset ? m->split.n_long : 0) : nullptr;
    if (a.n_rows == 0) return 0;
    int lpr = 64;
    while (lpr > 1 && lpr / 2 >= a.W4) lpr /= 2;
    if (lpr < 4) lpr = 4;
    const int rpw = 64 / lpr;
    const int waves = 4;
    auto blocks = [&](int64_t n) { return dim3((unsigned)((n + (int64_t)waves * rpw - 1) / (waves * rpw))); };
#define ELIMREC_HALF_LAUNCH(LPR)                                                                                     \
    do {                                                                                                             \
        const unsigned seg_blocks = has_split ? blocks(a.n_seg).x : 0u;                                              \
        if (a.src_mask)                                                                                              \
            hipLaunchKernelGGL((half_hop_kernel<LPR, 8, true>), dim3(seg_blocks + blocks(a.n_rows).x),               \
                               dim3(64 * waves), 0, s, a, (int)seg_blocks);                                          \
        else {                                                                                                       \
            /* streaming rows: a persistent grid (stream_wgs workgroups walk all items) */                           \
            unsigned row_blocks = blocks(a.row_list ? a.row_list_cap : a.n_rows).x;                                  \
            if (LPR >= 8 && a.row_items && !a.row_list && a.W4 <= LPR) {                                             \
                row_blocks = blocks(a.n_row_items).x;                                                                \
                if (row_blocks > (unsigned)stream_wgs()) row_blocks = (unsigned)stream_wgs();                        \
                if (row_blocks == 0) row_blocks = 1;                                                                 \
            }                                                                                                        \
            hipLaunchKernelGGL((half_hop_kernel<LPR, 8, false>), dim3(seg_blocks + row_blocks),                      \
                               dim3(64 * waves), 0, s, a, (int)seg_blocks);                                          \
        }                                                                                                            \
        ELIMREC_LAUNCH_CHECK("half_hop");                                                                            \
        if (has_split && !a.tickets) {                                                                               \
            hipLaunchKernelGGL((half_fixup_kernel<LPR>), dim3((unsigned)((a.n_long + waves - 1) / waves)), dim3(64 * waves), 0, s, a);               \
            ELIMREC_LAUNCH_CHECK("half_fixup");                                                                      \
        }                                                                                                            \
    } while (0)
    switch (lpr) {
        case 64: ELIMREC_HALF_LAUNCH(64); break;
        case 32: ELIMREC_HALF_LAUNCH(32); break;
        case 16: ELIMREC_HALF_LAUNCH(16); break;
        case 8: ELIMREC_HALF_LAUNCH(8); break;
        default: ELIMREC_HALF_LAUNCH(4); break;
    }
#undef ELIMREC_HALF_LAUNCH
    return 0;
}

static HalfArgs half_args(int W4, const float *Xin, const uint32_t *src_mask, float *Xout, const float *Add1,
                          const uint32_t *add1_mask, const float *Add2, const float *AddN, int N4, float *AccOut,
                          float scale) {
    HalfArgs a = {};
    a.W4 = W4; a.ld4 = W4; a.ld_add1 = W4; a.ld_acc = W4; a.Xin = (const float4 *)Xin; a.src_mask = src_mask; a.Xout = (float4 *)Xout;
    a.Add1 = (const float4 *)Add1; a.add1_mask = add1_mask; a.Add2 = (const float4 *)Add2;
    a.AddN = (const float4 *)AddN; a.N4 = N4 > 0 ? N4 : 1; a.AccOut = (float4 *)AccOut; a.scale = scale;
    return a;
}

}  // namespace elimrec

// The narrow (d-column) chain is independent of the wide chain for most of a propagation; it runs on a
// side stream forked from / joined to the caller's stream with events (also valid under stream capture).
// The partial-sum scratch of a split CSR holds two regions so both chains can use the same block at once:
// wide partials at offset 0, narrow partials at n_seg * C floats (callers allocate [n_seg x 2C]).
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
static SideStream &side_stream() {
    static thread_local SideStream s;
    if (!s.stream) {
        (void)hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        (void)hipEventCreateWithFlags(&s.fork, hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&s.join, hipEventDisableTiming);
    }
    return s;
}
static inline size_t narrow_partials_offset(const elimrec_csr *m, int C) { return (size_t)m->split.n_seg * C; }

static int g_ticket_fixup = 1;
extern "C" int elimrec_ticket_fixup(void) { return g_ticket_fixup; }
extern "C" void elimrec_set_ticket_fixup(int on) { g_ticket_fixup = on ? 1 : 0; }

static int g_concurrency = 0;                            // measured: +0.6 % at the Tiktok shape, so off by default
extern "C" int elimrec_concurrency(void) { return g_concurrency; }
extern "C" void elimrec_set_concurrency(int on) { g_concurrency = on ? 1 : 0; }

static int bwd_narrow_chain(const elimrec_csr *PT, const elimrec_csr *QT, int L, int C, int d4, const float *H_u,
                            const float *H_i, const uint32_t *mask_u, const uint32_t *mask_i, float *au, float *ai,
                            float *d_gEu, float inv, hipStream_t ns);

static size_t bip_ws_layout(int64_t U, int64_t I, int d, int C, size_t off[8]) {
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    off[0] = take((size_t)U * C * 4);   // wu
    off[1] = take((size_t)I * C * 4);   // wi
    off[2] = take((size_t)U * d * 4);   // au
    off[3] = take((size_t)I * d * 4);   // ai
    off[4] = take((size_t)U * d * 4);   // SN_u
    off[5] = take((size_t)I * d * 4);   // SN_i
    off[6] = take((size_t)(U + I) * d * 4);   // H (backward)
    off[7] = take((size_t)(((U + 31) / 32) + ((I + 31) / 32) + 2) * 4);   // masks
    return o;
}

extern "C" size_t elimrec_bipartite_workspace(int64_t U, int64_t I, int d, int M) {
    size_t off[8];
    return bip_ws_layout(U, I, d, d * M, off);
}

extern "C" int elimrec_propagate_bipartite(const elimrec_csr *P, const elimrec_csr *Q, int64_t U, int64_t I, int d,
                                           int M, int L, const float *d_user_emb, const float *d_XI, float *d_Out,
                                           float *d_narrow_out, void *d_workspace, size_t workspace_bytes,
                                           void *stream) {
    ELIMREC_REQUIRE(P && Q && d_user_emb && d_XI && d_Out && d_workspace, "propagate_bipartite: null pointer");
    ELIMREC_REQUIRE(d > 0 && d % 4 == 0 && M >= 1 && L >= 1, "propagate_bipartite: need d % 4 == 0, M >= 1, L >= 1");
    ELIMREC_REQUIRE(P->n_rows == U && Q->n_rows == I, "propagate_bipartite: block shapes do not match U, I");
    const int C = d * M;
    size_t off[8];
    if (workspace_bytes < bip_ws_layout(U, I, d, C, off)) {
        set_error("propagate_bipartite: workspace too small");
        return ELIMREC_E_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    float *wu = (float *)(ws + off[0]), *wi = (float *)(ws + off[1]);
    float *au = (float *)(ws + off[2]), *ai = (float *)(ws + off[3]);
    float *SNu = (float *)(ws + off[4]), *SNi = (float *)(ws + off[5]);
    float *Out_u = d_Out, *Out_i = d_Out + (size_t)U * C;
    const float inv = 1.0f / (float)(L + 1);
    const int C4 = C / 4, d4 = d / 4;
    int rc;
    // fork: the narrow chain runs on the side stream while the first wide step runs on the caller's stream
    SideStream &side = side_stream();
    const bool use_side = (elimrec_concurrency() != 0);
    hipStream_t ns = use_side ? side.stream : s;
    if (use_side) {
        if ((rc = check_hip(hipEventRecord(side.fork, s), "eventRecord(fork)"))) return rc;
        if ((rc = check_hip(hipStreamWaitEvent(ns, side.fork, 0), "streamWaitEvent(fork)"))) return rc;
    }
    // ---- narrow chain: a_k = A^k [E_u ; 0]; SN_side = sum of the a_k living on that side (k = 0 included)
    const float *prev = d_user_emb;
    for (int k = 1; k <= L; ++k) {
        const bool to_items = (k & 1);
        const bool need_next = k < L;
        if (to_items) {
            // first items step writes SN_i directly (it doubles as a_1); later ones accumulate in place
            float *xout = (k == 1) ? SNi : (need_next ? ai : nullptr);
            HalfArgs a = half_args(d4, prev, nullptr, xout, (k == 1) ? nullptr : SNi, nullptr, nullptr, nullptr, 0,
                                   (k == 1) ? nullptr : SNi, 1.0f);
            if ((rc = launch_half(Q, a, narrow_partials_offset(Q, C), ns))) return rc;
            prev = (k == 1) ? SNi : ai;
        } else {
            HalfArgs a = half_args(d4, prev, nullptr, need_next ? au : nullptr, (k == 2) ? d_user_emb : SNu, nullptr,
                                   nullptr, nullptr, 0, SNu, 1.0f);
            if ((rc = launch_half(P, a, narrow_partials_offset(P, C), ns))) return rc;
            prev = au;
        }
    }
    if (use_side && (rc = check_hip(hipEventRecord(side.join, ns), "eventRecord(join)"))) return rc;
    bool joined = !use_side;
    const float *SNu_final = (L >= 2) ? SNu : d_user_emb;     // users: a_0 = E_u (+ a_2 + ...)
    // ---- wide chain: w_k = A^k [0 ; XI]; running sums live in Out, the last step on a side finishes it
    const float *wprev = d_XI;
    bool users_done = false, items_done = false;
    for (int k = 1; k <= L; ++k) {
        const bool to_users = (k & 1);
        const bool last_on_side = (k + 2 > L);
        const bool need_next = k < L;
        if (!joined && (last_on_side || k >= 2)) {      // the narrow sums are consumed from here on
            if ((rc = check_hip(hipStreamWaitEvent(s, side.join, 0), "streamWaitEvent(join)"))) return rc;
            joined = true;
        }
        if (to_users) {
            const bool first = (k == 1);
            HalfArgs a = half_args(C4, wprev, nullptr, need_next ? wu : nullptr, first ? nullptr : Out_u, nullptr, nullptr,
                                   last_on_side ? SNu_final : nullptr, d4, Out_u, last_on_side ? inv : 1.0f);
            if ((rc = launch_half(P, a, 0, s))) return rc;
            wprev = wu;
            users_done = last_on_side;
        } else {
            const bool first = (k == 2);
            HalfArgs a = half_args(C4, wprev, nullptr, need_next ? wi : nullptr, first ? d_XI : Out_i, nullptr, nullptr,
                                   last_on_side ? SNi : nullptr, d4, Out_i, last_on_side ? inv : 1.0f);
            if ((rc = launch_half(Q, a, 0, s))) return rc;
            wprev = wi;
            items_done = last_on_side;
        }
    }
    (void)users_done;
    if (!joined && (rc = check_hip(hipStreamWaitEvent(s, side.join, 0), "streamWaitEvent(join)"))) return rc;
    if (d_narrow_out) {  // the shared narrow part of Out on both sides, scaled like Out
        hipLaunchKernelGGL(combine_kernel, dim3(1024), dim3(256), 0, s, (const float4 *)SNu_final, (const float4 *)nullptr,
                           U, d4, d4, inv, (float4 *)d_narrow_out);
        ELIMREC_LAUNCH_CHECK("combine(narrow users)");
        hipLaunchKernelGGL(combine_kernel, dim3(1024), dim3(256), 0, s, (const float4 *)SNi, (const float4 *)nullptr, I,
                           d4, d4, inv, (float4 *)(d_narrow_out + (size_t)U * d));
        ELIMREC_LAUNCH_CHECK("combine(narrow items)");
    }
    if (!items_done) {   // L == 1: Out_i = (XI + bcast(a_1)) / 2
        hipLaunchKernelGGL(combine_kernel, dim3(2048), dim3(256), 0, s, (const float4 *)d_XI, (const float4 *)SNi, I, C4,
                           d4, inv, (float4 *)Out_i);
        ELIMREC_LAUNCH_CHECK("combine");
    }
    return 0;
}

extern "C" int elimrec_propagate_bipartite_bwd(const elimrec_csr *PT, const elimrec_csr *QT, int64_t U, int64_t I,
                                               int d, int M, int L, const float *d_G, const float *d_H,
                                               const int32_t *d_active_rows, const int32_t *d_seg_info, int64_t n_max,
                                               float *d_gXI, float *d_gEu, void *d_workspace, size_t workspace_bytes,
                                               void *stream) {
    ELIMREC_REQUIRE(PT && QT && d_G && d_H && d_active_rows && d_seg_info && d_gXI && d_gEu && d_workspace,
                    "propagate_bipartite_bwd: null pointer");
    ELIMREC_REQUIRE(d > 0 && d % 4 == 0 && M >= 1 && L >= 1, "propagate_bipartite_bwd: need d % 4 == 0, M >= 1, L >= 1");
    ELIMREC_REQUIRE(PT->n_rows == I && QT->n_rows == U, "propagate_bipartite_bwd: block shapes do not match U, I");
    const int C = d * M;
    size_t off[8];
    if (workspace_bytes < bip_ws_layout(U, I, d, C, off)) {
        set_error("propagate_bipartite_bwd: workspace too small");
        return ELIMREC_E_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    float *wu = (float *)(ws + off[0]), *wi = (float *)(ws + off[1]);
    float *au = (float *)(ws + off[2]), *ai = (float *)(ws + off[3]);
    uint32_t *mask_u = (uint32_t *)(ws + off[7]);
    uint32_t *mask_i = mask_u + (U + 31) / 32 + 1;
    const size_t mask_bytes = (size_t)(((U + 31) / 32) + ((I + 31) / 32) + 2) * 4;
    int rc = check_hip(hipMemsetAsync(mask_u, 0, mask_bytes, s), "memset(masks)");
    if (rc) return rc;
    if (n_max > 0) {
        hipLaunchKernelGGL(build_masks_kernel, dim3((unsigned)((n_max + 255) / 256)), dim3(256), 0, s, d_active_rows,
                           d_seg_info, n_max, U, mask_u, mask_i);
        ELIMREC_LAUNCH_CHECK("build_masks");
    }
    const float inv = 1.0f / (float)(L + 1);
    const int C4 = C / 4, d4 = d / 4;
    const float *G_u = d_G, *G_i = d_G + (size_t)U * C;
    const float *H_u = d_H, *H_i = d_H + (size_t)U * d;
    // fork: narrow adjoint on the side stream, wide adjoint on the caller's stream
    SideStream &side = side_stream();
    const bool use_side = (elimrec_concurrency() != 0);
    hipStream_t ns = use_side ? side.stream : s;
    if (use_side) {
        if ((rc = check_hip(hipEventRecord(side.fork, s), "eventRecord(fork)"))) return rc;
        if ((rc = check_hip(hipStreamWaitEvent(ns, side.fork, 0), "streamWaitEvent(fork)"))) return rc;
    }
    if ((rc = bwd_narrow_chain(PT, QT, L, C, d4, H_u, H_i, mask_u, mask_i, au, ai, d_gEu, inv, ns))) return rc;
    // ---- wide adjoint: t_L = G_{s(L)}, t_k = G_{s(k)} + B_k t_{k+1}; s(k) = users for odd k. gXI = inv * t_0.
    {
        const float *t = (L & 1) ? G_u : G_i;
        const uint32_t *tmask = (L & 1) ? mask_u : mask_i;     // the raw gradient is row-sparse
        for (int k = L - 1; k >= 0; --k) {
            const bool out_items = !(k & 1);
            float *dst = (k == 0) ? d_gXI : (out_items ? wi : wu);
            HalfArgs a = half_args(C4, t, tmask, nullptr, out_items ? G_i : G_u, out_items ? mask_i : mask_u, nullptr,
                                   nullptr, 0, dst, (k == 0) ? inv : 1.0f);
            if ((rc = launch_half(out_items ? PT : QT, a, 0, s))) return rc;
            t = dst;
            tmask = nullptr;
        }
    }
    // ---- narrow adjoint: s'(k) = users for even k. gE_u = inv * t_0. Independent of the wide chain.
    if (use_side) {
        if ((rc = check_hip(hipEventRecord(side.join, ns), "eventRecord(join)"))) return rc;
        if ((rc = check_hip(hipStreamWaitEvent(s, side.join, 0), "streamWaitEvent(join)"))) return rc;
    }
    return 0;
}

static int bwd_narrow_chain(const elimrec_csr *PT, const elimrec_csr *QT, int L, int C, int d4, const float *H_u,
                            const float *H_i, const uint32_t *mask_u, const uint32_t *mask_i, float *au, float *ai,
                            float *d_gEu, float inv, hipStream_t ns) {
    int rc;
    {
        const float *t = (L & 1) ? H_i : H_u;
        const uint32_t *tmask = (L & 1) ? mask_i : mask_u;
        for (int k = L - 1; k >= 0; --k) {
            const bool out_users = !(k & 1);
            float *dst = (k == 0) ? d_gEu : (out_users ? au : ai);
            HalfArgs a = half_args(d4, t, tmask, nullptr, out_users ? H_u : H_i, out_users ? mask_u : mask_i, nullptr,
                                   nullptr, 0, dst, (k == 0) ? inv : 1.0f);
            const elimrec_csr *blk = out_users ? QT : PT;
            if ((rc = launch_half(blk, a, narrow_partials_offset(blk, C), ns))) return rc;
            t = dst;
            tmask = nullptr;
        }
    }
    return 0;
}

namespace elimrec {
__global__ __launch_bounds__(256) void blocksum_rows_kernel(const float *__restrict__ G,
                                                            const int32_t *__restrict__ active_rows,
                                                            const int32_t *__restrict__ seg_info, int64_t n_max, int d,
                                                            int M, int slot_major, float *__restrict__ H) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= n_max || s >= seg_info[0]) return;
    const int64_t r = active_rows[s];
    const float4 *g = reinterpret_cast<const float4 *>(G + (slot_major ? s : r) * (int64_t)d * M);
    float4 *h = reinterpret_cast<float4 *>(H + r * (int64_t)d);
    const int d4 = d / 4;
    for (int c = lane; c < d4; c += 64) {
        float4 acc = g[c];
        for (int m = 1; m < M; ++m) {
            const float4 x = g[m * d4 + c];
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
        h[c] = acc;
    }
}

__global__ void copy_cols_kernel(const float *__restrict__ src, int64_t ld_src, float *__restrict__ dst,
                                 int64_t ld_dst, int64_t n_rows, int n4) {
    const int64_t total = n_rows * n4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / n4;
        const int c = (int)(t - r * n4);
        *reinterpret_cast<float4 *>(dst + r * ld_dst + 4 * c) = *reinterpret_cast<const float4 *>(src + r * ld_src + 4 * c);
    }
}
}  // namespace elimrec

extern "C" int elimrec_blocksum_rows(const float *d_G, const int32_t *d_active_rows, const int32_t *d_seg_info,
                                     int64_t n_max, int d, int M, int slot_major, float *d_H, void *stream) {
    ELIMREC_REQUIRE(d_G && d_active_rows && d_seg_info && d_H, "blocksum_rows: null pointer");
    ELIMREC_REQUIRE(d > 0 && d % 4 == 0 && M >= 1, "blocksum_rows: bad d/M");
    if (n_max <= 0) return 0;
    hipLaunchKernelGGL(blocksum_rows_kernel, dim3((unsigned)((n_max + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_G,
                       d_active_rows, d_seg_info, n_max, d, M, slot_major, d_H);
    ELIMREC_LAUNCH_CHECK("blocksum_rows");
    return 0;
}

extern "C" int elimrec_copy_cols(const float *d_src, int64_t ld_src, float *d_dst, int64_t ld_dst, int64_t n_rows,
                                 int n_cols, void *stream) {
    ELIMREC_REQUIRE(d_src && d_dst, "copy_cols: null pointer");
    ELIMREC_REQUIRE(n_cols > 0 && n_cols % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0, "copy_cols: widths must be multiples of 4");
    if (n_rows <= 0) return 0;
    const int64_t total = n_rows * (n_cols / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_src, ld_src, d_dst,
                       ld_dst, n_rows, n_cols / 4);
    ELIMREC_LAUNCH_CHECK("copy_cols");
    return 0;
}

// Generic block SpMM with the fused epilogue on a column window of wider tables (leading dimension ld):
//   r = A . Xin[:, 0:W];  Xout = r;  AccOut = (r + Add1) * scale       (all [rows x W] windows, stride ld)
extern "C" int elimrec_block_spmm(const elimrec_csr *A, int W, int64_t ld, const float *d_Xin, float *d_Xout,
                                  const float *d_Add1, float *d_AccOut, float scale, void *stream) {
    ELIMREC_REQUIRE(A && d_Xin && (d_Xout || d_AccOut), "block_spmm: null pointer");
    ELIMREC_REQUIRE(W > 0 && W % 4 == 0 && ld % 4 == 0 && ld >= W, "block_spmm: W, ld must be multiples of 4, ld >= W");
    HalfArgs a = half_args(W / 4, d_Xin, nullptr, d_Xout, d_Add1, nullptr, nullptr, nullptr, 0, d_AccOut, scale);
    a.ld4 = a.ld_add1 = a.ld_acc = (int)(ld / 4);
    return launch_half(A, a, 0, (hipStream_t)stream);
}

// =====================================================================================================
// Folded propagation: with the constant feature tables folded into GEMM operands only ONE d-column table,
// X^0 = [E_u ; E_i], goes through the graph, and the two chains of the bipartite form (A^k [0;E_i] and
// A^k [E_u;0]) live on complementary sides of every layer -- so layer k of both is one ordinary hop
// X^k = A X^(k-1) over all N rows: ONE launch per hop. The epilogue keeps two sums:
//   Out0 = 1/(L+1) sum_k X^k                     (written into column block 0 of Out, row stride ldo)
//   Nar  = 1/(L+1) sum_{k even} X^k on user rows, sum_{k odd} X^k on item rows   (the part every table shares)
// Backward (adjoint, Horner): T^L = S^L, T^k = S^k + A^T T^(k+1), [gE_u ; gE_i] = 1/(L+1) T^0, where the source
// table S^k holds H (block sum of dOut) on the rows where the shared part lives at layer k and dOut block 0 on the
// others: SrcA = [H_u ; G_i] for even k, SrcB = [G_u ; H_i] for odd k (both valid on the active rows only:
// row bitmap). Requires an adjacency without diagonal blocks and L >= 1.
// =====================================================================================================
extern "C" size_t elimrec_folded_workspace(int64_t N, int d) {
    return 2 * align_up((size_t)N * d * sizeof(float), 256) + align_up((size_t)((N + 31) / 32 + 2) * 4, 256);
}

extern "C" int elimrec_propagate_folded(const elimrec_csr *A, int64_t U, int64_t I, int d, int L, const float *d_X0,
                                        float *d_Out0, int64_t ldo, float *d_narrow, void *d_workspace,
                                        size_t workspace_bytes, void *stream) {
    ELIMREC_REQUIRE(A && d_X0 && d_Out0 && d_narrow && d_workspace, "propagate_folded: null pointer");
    ELIMREC_REQUIRE(d > 0 && d % 4 == 0 && L >= 1 && ldo % 4 == 0 && ldo >= d, "propagate_folded: bad d/L/ldo");
    const int64_t N = U + I;
    ELIMREC_REQUIRE(A->n_rows == N, "propagate_folded: adjacency has %lld rows, expected %lld", (long long)A->n_rows, (long long)N);
    if (workspace_bytes < elimrec_folded_workspace(N, d)) { set_error("propagate_folded: workspace too small"); return ELIMREC_E_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    float *xb[2] = {(float *)ws, (float *)(ws + align_up((size_t)N * d * sizeof(float), 256))};
    const float inv = 1.0f / (float)(L + 1);
    const int d4 = d / 4;
    const float *xin = d_X0;
    int rc;
    for (int k = 1; k <= L; ++k) {
        const bool last = (k == L);
        float *xout = last ? nullptr : xb[(k - 1) & 1];
        HalfArgs a = half_args(d4, xin, nullptr, xout, (k == 1) ? d_X0 : d_Out0, nullptr, nullptr, nullptr, 0, d_Out0,
                               last ? inv : 1.0f);
        a.ld_add1 = (k == 1) ? d4 : (int)(ldo / 4);
        a.ld_acc = (int)(ldo / 4);
        // the shared (narrow) part lives on item rows for odd k, on user rows for even k
        const bool on_items = (k & 1);
        a.acc2_lo = on_items ? U : 0;
        a.acc2_hi = on_items ? N : U;
        a.Acc2Out = (float4 *)d_narrow;
        // first visit of a side: users start from a_0 = E_u (k = 2), items from nothing (k = 1)
        a.Acc2In = (k == 1) ? nullptr : ((k == 2) ? (const float4 *)d_X0 : (const float4 *)d_narrow);
        a.acc2_scale = (k + 2 > L) ? inv : 1.0f;          // last visit of this side
        if ((rc = launch_half(A, a, 0, s))) return rc;
        xin = xout;
    }
    if (L < 2) {      // the user rows of the shared part were never visited: Nar_u = a_0 / (L+1)
        hipLaunchKernelGGL(combine_kernel, dim3(1024), dim3(256), 0, s, (const float4 *)d_X0, (const float4 *)nullptr, U, d4,
                           d4, inv, (float4 *)d_narrow);
        ELIMREC_LAUNCH_CHECK("combine(narrow users)");
    }
    return 0;
}

namespace elimrec {
// SrcA[node] = node < U ? H : G ; SrcB[node] = node < U ? G : H, for the active nodes, from the slot-major dOut rows
// (G = column block 0, H = sum of the M blocks); also sets the row bitmap (pre-zeroed).
__global__ __launch_bounds__(256) void folded_sources_kernel(const float *__restrict__ dOutR,
                                                             const int32_t *__restrict__ active_rows,
                                                             const int32_t *__restrict__ seg_info, int64_t n_max,
                                                             int64_t U, int d, int M, float *__restrict__ SrcA,
                                                             float *__restrict__ SrcB, uint32_t *__restrict__ mask) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= n_max || s >= seg_info[0]) return;
    const int64_t r = active_rows[s];
    const float4 *g = reinterpret_cast<const float4 *>(dOutR + s * (int64_t)d * M);
    const int d4 = d / 4;
    float4 *h_dst = reinterpret_cast<float4 *>((r < U ? SrcA : SrcB) + r * (int64_t)d);
    float4 *g_dst = reinterpret_cast<float4 *>((r < U ? SrcB : SrcA) + r * (int64_t)d);
    for (int c = lane; c < d4; c += 64) {
        const float4 g0 = g[c];
        float4 acc = g0;
        for (int m = 1; m < M; ++m) {
            const float4 x = g[m * d4 + c];
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
        h_dst[c] = acc;
        g_dst[c] = g0;
    }
    if (mask && lane == 0) atomicOr(&mask[r >> 5], 1u << (r & 31));
}
}  // namespace elimrec

namespace elimrec {
// [H | G] of every active row -- H = sum of the M column blocks of its dOut row (block order), G = block 0: all the adjoint
// propagation needs of a dOut row -- cut into the `world` column slices a column-sharded job sends to its peers:
// out[w][s] = [H[s][w*dl : (w+1)*dl] | G[s][w*dl : (w+1)*dl]], dl = d / world  (layout [world x n_max x 2*dl])
__global__ __launch_bounds__(256) void source_rows_split_kernel(const float *__restrict__ dOutR, const int32_t *__restrict__ count,
                                                                int64_t n_max, int d4, int M, int dl4, float *__restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int sub = threadIdx.x & 15;
    if (s >= n_max || s >= *count) return;
    const float4 *g = reinterpret_cast<const float4 *>(dOutR) + s * (int64_t)d4 * M;
    float4 *o = reinterpret_cast<float4 *>(out);
    for (int c = sub; c < d4; c += 16) {
        const float4 g0 = g[c];
        float4 h = g0;
        for (int m = 1; m < M; ++m) {
            const float4 x = g[m * d4 + c];
            h.x += x.x; h.y += x.y; h.z += x.z; h.w += x.w;
        }
        const int w = c / dl4, j = c - w * dl4;
        float4 *row = o + ((int64_t)w * n_max + s) * 2 * dl4;
        row[j] = h;
        row[dl4 + j] = g0;
    }
}
}  // namespace elimrec

extern "C" int elimrec_source_rows_split(const float *d_dOutR, const int32_t *d_count, int64_t n_max, int d, int M, int world,
                                         float *d_out, void *stream) {
    ELIMREC_REQUIRE(d_dOutR && d_count && d_out, "source_rows_split: null pointer");
    ELIMREC_REQUIRE(d > 0 && M >= 1 && world >= 1 && d % world == 0 && (d / world) % 4 == 0, "source_rows_split: bad d/M/world");
    if (n_max <= 0) return 0;
    hipLaunchKernelGGL(source_rows_split_kernel, dim3((unsigned)((n_max + 15) / 16)), dim3(256), 0, (hipStream_t)stream, d_dOutR,
                       d_count, n_max, d / 4, M, d / world / 4, d_out);
    ELIMREC_LAUNCH_CHECK("source_rows_split");
    return 0;
}

extern "C" int elimrec_propagate_folded_bwd(const elimrec_csr *AT, int64_t U, int64_t I, int d, int M, int L,
                                            const float *d_dOutR, const int32_t *d_active_rows,
                                            const int32_t *d_seg_info, int64_t n_max, float *d_SrcA, float *d_SrcB,
                                            float *d_grad /* [N x d] = [gE_u ; gE_i] */,
                                            const uint32_t *d_active_mask, void *d_workspace, size_t workspace_bytes,
                                            void *stream) {
    // d_dOutR == NULL: the source tables and the row bitmap are already in place
    ELIMREC_REQUIRE(AT && d_SrcA && d_SrcB && d_grad && d_workspace, "propagate_folded_bwd: null pointer");
    ELIMREC_REQUIRE(d_dOutR ? (d_active_rows && d_seg_info) : (d_active_mask != nullptr),
                    "propagate_folded_bwd: dOut rows need their row list; prefilled sources need their bitmap");
    ELIMREC_REQUIRE(d > 0 && d % 4 == 0 && L >= 1 && M >= 1, "propagate_folded_bwd: bad d/L/M");
    const int64_t N = U + I;
    ELIMREC_REQUIRE(AT->n_rows == N, "propagate_folded_bwd: adjacency rows mismatch");
    if (workspace_bytes < elimrec_folded_workspace(N, d)) { set_error("propagate_folded_bwd: workspace too small"); return ELIMREC_E_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    const size_t tb = align_up((size_t)N * d * sizeof(float), 256);
    float *tbuf[2] = {(float *)ws, (float *)(ws + tb)};
    // bitmap of the active rows: the caller's (elimrec_segment_plan builds it on the way) or built here
    uint32_t *own_mask = d_active_mask ? nullptr : (uint32_t *)(ws + 2 * tb);
    const uint32_t *mask = d_active_mask ? d_active_mask : own_mask;
    int rc = 0;
    if (own_mask && (rc = check_hip(hipMemsetAsync(own_mask, 0, (size_t)((N + 31) / 32 + 2) * 4, s), "memset(mask)"))) return rc;
    if (d_dOutR && n_max > 0) {
        hipLaunchKernelGGL(folded_sources_kernel, dim3((unsigned)((n_max + 3) / 4)), dim3(256), 0, s, d_dOutR, d_active_rows,
                           d_seg_info, n_max, U, d, M, d_SrcA, d_SrcB, own_mask);
        ELIMREC_LAUNCH_CHECK("folded_sources");
    }
    const float inv = 1.0f / (float)(L + 1);
    const int d4 = d / 4;
    const float *t = (L & 1) ? d_SrcB : d_SrcA;       // T^L = S^L (row-sparse)
    const uint32_t *tmask = mask;
    for (int k = L - 1; k >= 0; --k) {
        float *dst = (k == 0) ? d_grad : tbuf[k & 1];
        HalfArgs a = half_args(d4, t, tmask, nullptr, (k & 1) ? d_SrcB : d_SrcA, mask, nullptr, nullptr, 0, dst,
                               (k == 0) ? inv : 1.0f);
        if ((rc = launch_half(AT, a, 0, s))) return rc;
        t = dst;
        tmask = nullptr;
    }
    return 0;
}
