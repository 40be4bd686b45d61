// Layer means of the folded propagation at listed rows (elimrec_slab_rows; the reference's torch.stack(embs).mean over the
// L + 1 propagated tables, models/EliMRec.py:244-247, evaluated at the batch's rows only): the argument block and the
// per-(row, column) body, shared by slab_rows_kernel (slab.hip) and the head forward that evaluates its own rows (head.hip).
#pragma once
#include "common.h"

namespace elimrec {

constexpr int kSlabMaxLayers = 8;

struct RowsArgs {
    const float4 *x[kSlabMaxLayers + 1];
    int L;
    int64_t U, n_rows;
    int nc4, w4, w4_shift;
    const float4 *long_tab;
    int n_long;
    const int32_t *long_index, *col;
    const int64_t *rowptr;          // plain CSR row pointers: 64-bit (a 2e9-non-zero graph, BASELINE.json configs[4])
    const float *val;
    const int32_t *rows, *counts;
    int64_t R;
    int n_lists;
    float *out0;
    int64_t ld_out0;
    float *narrow;
    int64_t ld_narrow;
    int by_node;
    float inv;
};

// float4 column c of row r: out = 1/(L+1) * (((X^0 + X^1) + X^2) + ... + X^L), nar = 1/(L+1) * sum over even k (user rows) /
// odd k (item rows) of X^k. Hop L inline when its table is absent: X^L[r] = sum_j A[r, j] X^(L-1)[j] through the plain CSR,
// 16 neighbours in flight; split rows come out of long_tab.
__device__ __forceinline__ void rows_piece(const RowsArgs &a, int64_t r, int c, float4 &out, float4 &nar_out) {
    constexpr int U8 = 16;
    const bool user = r < a.U;
    // layer pointers by compare-select over constant indices: indexing the by-value argument array with a run-time k
    // makes the compiler spill it to scratch (32 B of private segment, and the scratch set-up with it)
    auto layer = [&](int k) -> const float4 * {
        const float4 *p = a.x[0];
#pragma unroll
        for (int q = 1; q <= kSlabMaxLayers; ++q) p = (k == q) ? a.x[q] : p;
        return p;
    };
    const float4 *xL = layer(a.L), *xLm1 = layer(a.L - 1);
    const bool inline_hop = xL == nullptr;
    int li = -1;
    int64_t beg = 0, end = 0;
    if (inline_hop) {
        li = a.long_index[r];
        if (li < 0) { beg = a.rowptr[r]; end = a.rowptr[r + 1]; }
    }
    const int slab = c >> a.w4_shift, c4 = c & (a.w4 - 1);
    const int64_t idx = ((int64_t)slab * a.n_rows + r) * a.w4 + c4;
    float4 xl;
    if (!inline_hop) xl = xL[idx];
    else if (li >= 0) xl = a.long_tab[((int64_t)slab * a.n_long + li) * a.w4 + c4];
    else {
        const float4 *X = xLm1 + (int64_t)slab * a.n_rows * a.w4 + c4;
        xl = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t j = beg; j < end; j += U8) {
            int cj[U8];
            float vj[U8];
            float4 x[U8];
#pragma unroll
            for (int u = 0; u < U8; ++u) {
                const bool in = (j + u) < end;
                cj[u] = in ? a.col[j + u] : 0;
                vj[u] = in ? a.val[j + u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U8; ++u)
                x[u] = (j + u) < end ? X[(int64_t)cj[u] * a.w4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < U8; ++u) {
                xl.x = fmaf(vj[u], x[u].x, xl.x); xl.y = fmaf(vj[u], x[u].y, xl.y);
                xl.z = fmaf(vj[u], x[u].z, xl.z); xl.w = fmaf(vj[u], x[u].w, xl.w);
            }
        }
    }
    const float4 x0 = a.x[0][idx];
    const float4 x1 = (a.L == 1) ? xl : a.x[1][idx];
    float4 sum = make_float4(x0.x + x1.x, x0.y + x1.y, x0.z + x1.z, x0.w + x1.w);
    float4 nar = user ? x0 : x1;
    for (int k = 2; k <= a.L; ++k) {
        const float4 v = (k == a.L) ? xl : layer(k)[idx];
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        if (((k & 1) == 0) == user) { nar.x += v.x; nar.y += v.y; nar.z += v.z; nar.w += v.w; }
    }
    out = make_float4(sum.x * a.inv, sum.y * a.inv, sum.z * a.inv, sum.w * a.inv);
    nar_out = make_float4(nar.x * a.inv, nar.y * a.inv, nar.z * a.inv, nar.w * a.inv);
}

// Host side: the argument block of a rows evaluation over plan A (geometry ns x w floats per row)
static inline int rows_args_fill(const char *who, const elimrec_sell *A, int ns, int w, int L, int64_t U, const float *const *layers,
                                 const float *d_long, RowsArgs &a) {
    if (!(A && layers)) { set_error("%s: null pointer", who); return ELIMREC_E_BADARG; }
    if (!(L >= 1 && L <= kSlabMaxLayers)) { set_error("%s: 1 <= L <= %d", who, kSlabMaxLayers); return ELIMREC_E_BADARG; }
    int w4_shift = -1;
    if (w > 0 && w % 4 == 0) { int v = w / 4; w4_shift = 0; while ((1 << w4_shift) < v) ++w4_shift; if ((1 << w4_shift) != v) w4_shift = -1; }
    if (A->n_rows < 0 || ns < 1 || w4_shift < 0) { set_error("%s: bad slab geometry (n=%lld, ns=%d, w=%d)", who, (long long)A->n_rows, ns, w); return ELIMREC_E_BADARG; }
    for (int k = 0; k <= L; ++k) a.x[k] = (const float4 *)layers[k];
    for (int k = 0; k < L; ++k)
        if (!layers[k]) { set_error("%s: layer table %d missing", who, k); return ELIMREC_E_BADARG; }
    if (!(layers[L] || (A->d_rowptr && A->d_csr_col && A->d_csr_val && A->d_long_index && (A->n_long == 0 || d_long)))) {
        set_error("%s: the inline last hop needs the CSR, the long-row index and the long-row table", who);
        return ELIMREC_E_BADARG;
    }
    a.L = L; a.U = U; a.n_rows = A->n_rows; a.nc4 = ns * (w / 4); a.w4 = w / 4; a.w4_shift = w4_shift;
    a.long_tab = (const float4 *)d_long; a.n_long = A->n_long; a.long_index = A->d_long_index;
    a.rowptr = A->d_rowptr; a.col = A->d_csr_col; a.val = A->d_csr_val;
    a.inv = 1.0f / (float)(L + 1);
    return 0;
}

}  // namespace elimrec
