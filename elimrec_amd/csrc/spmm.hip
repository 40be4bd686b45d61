// LightGCN propagation for all M tables at once (models/EliMRec.py:238-248, K2-K4 of SURVEY §2a).
//
// Layout: one [N x C] fp32 table, C = M*d; at the Tiktok shape a row is 1 KiB = one wave x
// float4, so a neighbour row is fetched by ONE fully coalesced wave instruction and one pass
// over the CSR structure serves every table. HBM-bound: per hop read C*4 B per non-zero
// (+ 8 B of CSR), write C*4 B per row.
#include "common.h"

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
