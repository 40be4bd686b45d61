// The propagation matrix on the device (models/EliMRec.py:309-354, create_adj_mat): from the unique training
// interactions (u, i) to the CSR of
//     plain   A = [[0, R], [R^T, 0]]
//     pre     D^-1/2 A D^-1/2                 (inf -> 0)
//     gcmc    D^-1 A
//     norm    D'^-1 (A + I),  D' = D + I
//     mean    D^-1 A + I                      (the reference's fall-through branch)
// with int32 row pointers / column indices sorted inside a row and fp32 values that are BIT-IDENTICAL to what scipy
// produces on the host: the only inexact step of the reference is d^p for the (small, integer) degrees, so the host
// hands over a table pow_table[k] = float32(numpy.power(float32(k), p)) for k = 0..max degree (+1) computed by numpy
// itself, and everything the device adds is single IEEE fp32 multiplications in scipy's order
// (D.dot(A).dot(D): (d_r * 1) * d_c) -- exact by construction, checked against create_adj_mat on every fixture.
//
// Pipeline: 64-bit keys (row << 32 | col) of both orientations (+ the diagonal) -> rocPRIM radix sort -> row pointers
// by binary search, columns = low words, degrees = row lengths without the diagonal, values from the table.
#include <cstring>
#include "common.h"
#include <rocprim/rocprim.hpp>

namespace elimrec {

__global__ void adj_keys_kernel(const int64_t *__restrict__ u, const int64_t *__restrict__ it, int64_t E, int64_t U, int64_t N,
                                int with_diag, uint64_t *__restrict__ keys) {
    const int64_t total = 2 * E + (with_diag ? N : 0);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        uint64_t r, c;
        if (e < E) { r = (uint64_t)u[e]; c = (uint64_t)(U + it[e]); }
        else if (e < 2 * E) { r = (uint64_t)(U + it[e - E]); c = (uint64_t)u[e - E]; }
        else { r = c = (uint64_t)(e - 2 * E); }
        keys[e] = (r << 32) | c;
    }
}

__global__ void adj_rowptr_kernel(const uint64_t *__restrict__ keys, int64_t nnz, int64_t N, int32_t *__restrict__ rowptr) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > N) return;
    const uint64_t want = (uint64_t)r << 32;            // first key of row r
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    rowptr[r] = (int32_t)lo;
}

// adj_type: 0 plain, 1 pre, 2 gcmc, 3 norm, 4 mean (+ I)
__global__ void adj_values_kernel(const uint64_t *__restrict__ keys, const int32_t *__restrict__ rowptr, int64_t nnz, int64_t N,
                                  int adj_type, int with_diag, const float *__restrict__ pow_table, int table_len,
                                  int32_t *__restrict__ col, float *__restrict__ val, int32_t *__restrict__ err) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[p];
        const int64_t r = (int64_t)(k >> 32), c = (int64_t)(k & 0xFFFFFFFFull);
        col[p] = (int32_t)c;
        if (p > 0 && keys[p - 1] == k) atomicOr(err, 1);          // duplicate interaction: scipy would sum it
        // degrees = row lengths of A (the diagonal entry of A + I is not a neighbour)
        const int dr = rowptr[r + 1] - rowptr[r] - (with_diag ? 1 : 0);
        const int dc = rowptr[c + 1] - rowptr[c] - (with_diag ? 1 : 0);
        float v = 1.0f;
        if (adj_type == 1) {                     // (d_r^-1/2 * 1) * d_c^-1/2
            if (dr >= table_len || dc >= table_len) { atomicOr(err, 2); continue; }
            v = (pow_table[dr] * 1.0f) * pow_table[dc];
        } else if (adj_type == 2) {              // d_r^-1 * 1
            if (dr >= table_len) { atomicOr(err, 2); continue; }
            v = pow_table[dr] * 1.0f;
        } else if (adj_type == 3) {              // (d_r + 1)^-1 * (A + I)_rc
            if (dr + 1 >= table_len) { atomicOr(err, 2); continue; }
            v = pow_table[dr + 1] * 1.0f;
        } else if (adj_type == 4) {              // d_r^-1 A + I
            if (dr >= table_len) { atomicOr(err, 2); continue; }
            v = (r == c) ? 1.0f : pow_table[dr] * 1.0f;
        }
        val[p] = v;
    }
}

}  // namespace elimrec

using namespace elimrec;

extern "C" size_t elimrec_build_adj_workspace(int64_t E, int64_t N, int with_diag) {
    const size_t n = (size_t)(2 * E + (with_diag ? N : 0));
    size_t sort_bytes = 0;
    (void)rocprim::radix_sort_keys<rocprim::default_config, const uint64_t *, uint64_t *>(nullptr, sort_bytes, nullptr, nullptr, n, 0, 64,
                                                                                        0, false);
    return align_up(n * sizeof(uint64_t), 256) * 2 + align_up(sort_bytes, 256) + 256;
}

extern "C" int elimrec_build_adj(const int64_t *d_users, const int64_t *d_items, int64_t E, int64_t U, int64_t I, int adj_type,
                                 const float *d_pow_table, int table_len, int32_t *d_rowptr, int32_t *d_col, float *d_val,
                                 int32_t *d_err, void *d_workspace, size_t workspace_bytes, void *stream) {
    ELIMREC_REQUIRE(d_users && d_items && d_rowptr && d_col && d_val && d_err && d_workspace, "build_adj: null pointer");
    ELIMREC_REQUIRE(adj_type >= 0 && adj_type <= 4, "build_adj: adj_type 0..4 (plain, pre, gcmc, norm, mean)");
    ELIMREC_REQUIRE(adj_type == 0 || (d_pow_table && table_len > 0), "build_adj: the degree-power table is missing");
    const int64_t N = U + I;
    ELIMREC_REQUIRE(E >= 0 && N > 0 && N < INT32_MAX && 2 * E + N < INT32_MAX, "build_adj: graph too large for int32 CSR");
    const int with_diag = adj_type >= 3;
    if (workspace_bytes < elimrec_build_adj_workspace(E, N, with_diag)) { set_error("build_adj: workspace too small"); return ELIMREC_E_WORKSPACE; }
    const size_t n = (size_t)(2 * E + (with_diag ? N : 0));
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)d_workspace;
    uint64_t *keys = (uint64_t *)ws, *sorted = (uint64_t *)(ws + align_up(n * sizeof(uint64_t), 256));
    void *tmp = ws + 2 * align_up(n * sizeof(uint64_t), 256);
    size_t sort_bytes = workspace_bytes - 2 * align_up(n * sizeof(uint64_t), 256);
    int rc = check_hip(hipMemsetAsync(d_err, 0, sizeof(int32_t), s), "memset(err)");
    if (rc) return rc;
    if (n > 0) {
        hipLaunchKernelGGL(adj_keys_kernel, dim3(2048), dim3(256), 0, s, d_users, d_items, E, U, N, with_diag, keys);
        ELIMREC_LAUNCH_CHECK("adj_keys");
        hipError_t e = rocprim::radix_sort_keys(tmp, sort_bytes, (const uint64_t *)keys, sorted, n, 0, 64, s, false);
        if (e != hipSuccess) return check_hip(e, "radix_sort_keys");
    }
    hipLaunchKernelGGL(adj_rowptr_kernel, dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, s, (const uint64_t *)sorted, (int64_t)n, N,
                       d_rowptr);
    ELIMREC_LAUNCH_CHECK("adj_rowptr");
    if (n > 0) {
        hipLaunchKernelGGL(adj_values_kernel, dim3(2048), dim3(256), 0, s, (const uint64_t *)sorted, (const int32_t *)d_rowptr, (int64_t)n,
                           N, adj_type, with_diag, d_pow_table, table_len, d_col, d_val, d_err);
        ELIMREC_LAUNCH_CHECK("adj_values");
    }
    return 0;
}
