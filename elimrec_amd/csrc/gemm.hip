// Dense projections of the EliMRec hot path on the gfx950 matrix cores, fp32 in / fp32 out.
//
//   linear_fwd   C = A . W^T + b      (K1 feature projection, K5 fusion Linear, K8 s_dense)
//   linear_bwd_w out = A^T . B        (their weight gradients), deterministic split over rows
//
// v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain (one rounding per product), so results
// stay within fp32 round-off of the reference's addmm.
//
// Operand maps (cdna_hip_programming.md §3): for D[32x32] += A[32x2] . B[2x32]
//   lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
//   accumulator register r of lane l is D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31].
#include "common.h"
#include "merge_rows.h"
#include "bwd_w.h"
#include <cstdlib>

namespace elimrec {

typedef float v16f __attribute__((ext_vector_type(16)));

// --------------------------------------------------------------------------------- forward
// Workgroup = 4 waves, tile 128 rows x 64 cols, K staged 32 at a time through LDS
// (row stride 33 floats: the 32 lanes of a half-wave read 32 different rows at the same k).
constexpr int FBN = 64, FBK = 16, FLD = FBK + 1;

struct FwdBatch { elimrec_linear_desc p[kMaxBatch]; };

// blockIdx.z selects the problem: independent Linears (the three feature projections; the five head
// Linears) share one launch so the grid fills the chip.
// BM = 128: wave w owns rows [32w, 32w+32) x all 64 columns (two MFMA tiles share the A operand).
// BM = 64 : wave w owns rows [32(w&1), +32) x columns [32(w>>1), +32) (one tile); twice the workgroups,
//           smaller LDS footprint -- better when the grid is only a couple of rounds deep.
template <int BM, int PD>
__global__ __launch_bounds__(256) void linear_fwd_kernel(FwdBatch batch) {
    const elimrec_linear_desc &pd = batch.p[blockIdx.z];
    const float *__restrict__ A = pd.d_A;
    const float *__restrict__ W = pd.d_W;
    const float *__restrict__ bias = pd.d_bias;
    float *__restrict__ C = pd.d_C;
    const float *__restrict__ rowscale = pd.d_rowscale;
    const float *__restrict__ addm = pd.d_add;
    const int32_t *__restrict__ row_index = pd.d_row_index;   // A / rowscale / add row of output row m
    const int64_t lda = pd.lda, ldw = pd.ldw, ldc = pd.ldc, ldadd = pd.ldadd;
    // output rows [row_lo, M): the whole problem, or the device-side range (active rows of this batch)
    const int64_t row_lo = pd.d_row_range ? (int64_t)max(pd.d_row_range[0], 0) : 0;
    const int64_t M = pd.d_row_range ? (int64_t)min((int64_t)pd.d_row_range[1], pd.M) : pd.M;
    const int N = pd.N, K = pd.K;
    const bool relu = pd.act == 1;
    const int n0 = blockIdx.y * FBN;
    const int64_t n_tiles = M > row_lo ? (M - row_lo + BM - 1) / BM : 0;
    if ((int64_t)blockIdx.x >= n_tiles || n0 >= N) return;
    // PERSISTENT over row tiles: workgroup b takes tiles b, b + gridDim.x, ...; the (tile, K-chunk) pairs form
    // one software pipeline -- while chunk g feeds the MFMAs, chunk g+1 (possibly the first chunk of the NEXT
    // tile) is in flight in registers, so the global-load latency is paid once per workgroup, not per tile.
    __shared__ float As[2][BM * FLD];
    __shared__ float Bs[2][FBN * FLD];
    constexpr int KP4 = FBK / 4;             // float4 per row of a K-chunk
    constexpr int RPP = 256 / KP4;           // rows covered by one pass of the 256 loader threads
    constexpr int AP = BM / RPP;             // float4 loads per thread for the A chunk
    constexpr int BP = FBN / RPP;            // ... for the W chunk
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lr = tid / KP4;           // row inside a pass
    const int lc = (tid % KP4) * 4;     // k offset of this thread's float4
    const int wrow = (BM == 128) ? wave * 32 : (wave & 1) * 32;
    const int wcol = (BM == 128) ? 0 : (wave >> 1) * 32;
    const int chunks = (K + FBK - 1) / FBK;
    const int64_t my_tiles = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int64_t G = my_tiles * chunks;

    v16f acc0 = {0}, acc1 = {0};
    const int ai = lane & 31, ak = lane >> 5;
    // PD K-chunks are in flight in registers at any time. One chunk is only 0.25 us of MFMA work against a
    // ~2 us global load: with many resident workgroups (large M) other waves hide that and PD = 1 (least
    // registers, most waves) is fastest; a launch of a few hundred tiles (the batch-row projections) has one
    // workgroup per CU and runs at the latency of its own chain, so it wants the deep pipeline.
    float4 ra[PD][AP], rb[PD][BP];
    auto load_chunk = [&](int64_t g, float4 (&qa)[AP], float4 (&qb)[BP]) {
        const int64_t m0 = row_lo + ((int64_t)blockIdx.x + (g / chunks) * gridDim.x) * BM;
        const int k0 = (int)(g % chunks) * FBK;
        const bool kin = (k0 + lc) < K;  // K % 4 == 0: the whole float4 is in or out
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int64_t gr = m0 + lr + RPP * i;
            const int64_t sr = (row_index && gr < M) ? (int64_t)row_index[gr] : gr;
            qa[i] = (kin && gr < M) ? *reinterpret_cast<const float4 *>(A + sr * lda + k0 + lc)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int gn = n0 + lr + RPP * i;
            qb[i] = (kin && gn < N) ? *reinterpret_cast<const float4 *>(W + (int64_t)gn * ldw + k0 + lc)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_chunk = [&](int buf, const float4 (&qa)[AP], const float4 (&qb)[BP]) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            float *dst = &As[buf][(lr + RPP * i) * FLD + lc];
            dst[0] = qa[i].x; dst[1] = qa[i].y; dst[2] = qa[i].z; dst[3] = qa[i].w;
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            float *dst = &Bs[buf][(lr + RPP * i) * FLD + lc];
            dst[0] = qb[i].x; dst[1] = qb[i].y; dst[2] = qb[i].z; dst[3] = qb[i].w;
        }
    };
    const int col0 = n0 + wcol + (lane & 31);
    const int col1 = col0 + 32;
    const float bias0 = (bias && col0 < N) ? bias[col0] : 0.f;
    const float bias1 = (BM == 128 && bias && col1 < N) ? bias[col1] : 0.f;

#pragma unroll
    for (int j = 0; j < PD; ++j)
        if (j < G) load_chunk(j, ra[j], rb[j]);
    int buf = 0;
    for (int64_t g0 = 0; g0 < G; g0 += PD) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            const int64_t g = g0 + j;
            if (g >= G) break;
            // double-buffered LDS, one barrier per chunk: a wave that stores into `buf` here has passed the
            // barrier of chunk g-1, which every wave reaches only after its MFMAs of chunk g-2 (same buffer)
            store_chunk(buf, ra[j], rb[j]);
            __syncthreads();
            if (g + PD < G) load_chunk(g + PD, ra[j], rb[j]);
            const float *ap = &As[buf][(wrow + ai) * FLD + ak];
            const float *bp0 = &Bs[buf][(wcol + ai) * FLD + ak];
            const float *bp1 = &Bs[buf][(32 + ai) * FLD + ak];
#pragma unroll
            for (int kk = 0; kk < FBK; kk += 2) {
                const float a = ap[kk];
                const float b0 = bp0[kk];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
                if (BM == 128) {
                    const float b1 = bp1[kk];
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
                }
            }
            if ((int)(g % chunks) == chunks - 1) {       // last K-chunk of this tile: write it out, start the next
                const int64_t m0 = row_lo + ((int64_t)blockIdx.x + (g / chunks) * gridDim.x) * BM;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = m0 + wrow + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < M) {
                        const int64_t sr = row_index ? (int64_t)row_index[row] : row;
                        const float rs = rowscale ? rowscale[sr] : 1.f;      // bias enters as rowscale[row] * bias[n]
                        if (col0 < N) {
                            const float v = acc0[r] + rs * bias0 + (addm ? addm[sr * ldadd + col0] : 0.f);
                            C[row * ldc + col0] = relu ? fmaxf(v, 0.f) : v;
                        }
                        if (BM == 128 && col1 < N) {
                            const float v = acc1[r] + rs * bias1 + (addm ? addm[sr * ldadd + col1] : 0.f);
                            C[row * ldc + col1] = relu ? fmaxf(v, 0.f) : v;
                        }
                    }
                }
                acc0 = (v16f){0};
                acc1 = (v16f){0};
            }
            buf ^= 1;
        }
    }
}

// --------------------------------------------------------------------------------- weight grad
// out[i, j] = sum_r A[r, i] * B[idx(r), j]   (A^T . B; the reduction runs over ROWS).
// A workgroup owns one chunk of rows and one 64 x TN2 output tile (4 waves x TN2/64 MFMA tiles).
// Rows are staged 32 at a time through LDS with 16-byte loads (a row of A or B is contiguous),
// double-buffered: the global loads of block t+1 are in flight while block t feeds the MFMAs.
// MFMA operands come straight from the row-major LDS image: lane (i, k) of the A operand is
// As[k][i0 + i] -- consecutive lanes, consecutive addresses, no transpose anywhere.
// Each workgroup writes its partial tile to slab[chunk]; slabs are summed in chunk order
// (reduce_slabs_kernel), so the result is bitwise reproducible.
__global__ __launch_bounds__(256) void linear_bwd_w_partial_kernel(BwdBatch batch) {
    __shared__ float As[2][TRB * TN1];
    __shared__ float Bs[2][TRB * TN2];
    bwd_w_partial_body(batch, (int)blockIdx.x, As, Bs);
}

// The same launch with the adjoint-source merge (merge_rows.h) as `merge_blocks` extra workgroups, in front of
// (merge_first) or behind the weight-gradient workgroups: both read the head backward's dOut rows and nothing of each
// other, and the merge alone is a latency-bound launch of its own on the step's critical path.
__global__ __launch_bounds__(256) void linear_bwd_w_merge_kernel(BwdBatch batch, MergeArgs mg, int bw_blocks, int merge_blocks,
                                                                 int merge_first) {
    __shared__ float As[2][TRB * TN1];
    __shared__ float Bs[2][TRB * TN2];
    int b = (int)blockIdx.x;
    bool merge;
    if (merge_first) { merge = b < merge_blocks; if (!merge) b -= merge_blocks; }
    else { merge = b >= bw_blocks; if (merge) b -= bw_blocks; }
    if (merge) {
        // LDS of the GEMM stages, reused: two rank tables + the chunk's row bits
        int *s_beg = reinterpret_cast<int *>(&Bs[0][0]);
        slab_merge_rows_body(mg, b, s_beg, s_beg + kSlabMaxRanks, reinterpret_cast<uint32_t *>(&As[0][0]));
        return;
    }
    bwd_w_partial_body(batch, b, As, Bs);
}

__global__ void reduce_slabs_kernel(BwdBatch batch) { reduce_slabs_body(batch, (int)blockIdx.x, (int)blockIdx.y); }

}  // namespace elimrec

using namespace elimrec;

extern "C" int elimrec_linear_fwd_batched(const elimrec_linear_desc *descs, int n, void *stream) {
    ELIMREC_REQUIRE(descs && n >= 1 && n <= kMaxBatch, "linear_fwd_batched: 1..%d problems", kMaxBatch);
    FwdBatch batch;
    int64_t max_tiles_m = 0;
    int max_tiles_n = 0;
    for (int i = 0; i < n; ++i) {
        const elimrec_linear_desc &d = descs[i];
        ELIMREC_REQUIRE(d.d_A && d.d_W && d.d_C, "linear_fwd: null pointer");
        ELIMREC_REQUIRE(d.M >= 0 && d.N > 0 && d.K > 0, "linear_fwd: bad shape M=%lld N=%d K=%d", (long long)d.M, d.N, d.K);
        ELIMREC_REQUIRE(d.K % 4 == 0 && d.lda % 4 == 0 && d.ldw % 4 == 0, "linear_fwd: K, lda, ldw must be multiples of 4");
        ELIMREC_REQUIRE(d.act == 0 || d.act == 1, "linear_fwd: unknown activation %d", d.act);
        ELIMREC_REQUIRE(((uintptr_t)d.d_A % 16) == 0 && ((uintptr_t)d.d_W % 16) == 0, "linear_fwd: A and W must be 16-byte aligned");
        batch.p[i] = d;
        if (d.M > max_tiles_m) max_tiles_m = d.M;       // rows; converted to tiles below
        const int tn = (d.N + FBN - 1) / FBN;
        if (tn > max_tiles_n) max_tiles_n = tn;
    }
    if (max_tiles_m == 0) return 0;
    const int tile_rows = 64;                     // (128-row tiles measured slower at these shapes)
    const int wg_budget = 1 << 30;                // one workgroup per tile (measured best; a smaller budget makes the kernel persistent)
    int64_t per_problem = (wg_budget + (int64_t)n * max_tiles_n - 1) / ((int64_t)n * max_tiles_n);
    if (tile_rows == 128) {
        const int64_t tiles = (max_tiles_m + 127) / 128;
        dim3 grid((unsigned)(tiles < per_problem ? tiles : per_problem), (unsigned)max_tiles_n, (unsigned)n);
        hipLaunchKernelGGL((linear_fwd_kernel<128, 1>), grid, dim3(256), 0, (hipStream_t)stream, batch);
    } else {
        const int64_t tiles = (max_tiles_m + 63) / 64;
        dim3 grid((unsigned)(tiles < per_problem ? tiles : per_problem), (unsigned)max_tiles_n, (unsigned)n);
        const int depth = 0;
        // small launches (about one workgroup per CU or less) take the deep software pipeline
        const int64_t wgs = (int64_t)grid.x * grid.y * grid.z;
        const int pd = depth > 0 ? depth : (wgs <= 768 ? 8 : 1);
        if (pd >= 8) hipLaunchKernelGGL((linear_fwd_kernel<64, 8>), grid, dim3(256), 0, (hipStream_t)stream, batch);
        else if (pd >= 4) hipLaunchKernelGGL((linear_fwd_kernel<64, 4>), grid, dim3(256), 0, (hipStream_t)stream, batch);
        else hipLaunchKernelGGL((linear_fwd_kernel<64, 1>), grid, dim3(256), 0, (hipStream_t)stream, batch);
    }
    ELIMREC_LAUNCH_CHECK("linear_fwd");
    return 0;
}

extern "C" int elimrec_linear_fwd(const float *d_A, int64_t lda, const float *d_W, int64_t ldw,
                                  const float *d_bias, float *d_C, int64_t ldc, int64_t M, int N, int K,
                                  void *stream) {
    elimrec_linear_desc d = {d_A, lda, d_W, ldw, d_bias, d_C, ldc, M, N, K, nullptr, nullptr, 0, nullptr, nullptr, 0};
    return elimrec_linear_fwd_batched(&d, 1, stream);
}

extern "C" size_t elimrec_linear_bwd_w_workspace(int64_t R, int n1, int n2) { return bwd_w_bytes(R, n1, n2); }

extern "C" size_t elimrec_linear_bwd_w_batched_workspace(const elimrec_linear_bwd_desc *descs, int n) {
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += bwd_w_bytes(descs[i].R, descs[i].n1, descs[i].n2);
    return total;
}

static int linear_bwd_w_batched_impl(const elimrec_linear_bwd_desc *descs, int n, void *d_workspace, size_t workspace_bytes,
                                     const MergeArgs *mg, int defer_reduce, void *stream) {
    ELIMREC_REQUIRE(descs && n >= 1 && n <= kMaxBatch, "linear_bwd_w_batched: 1..%d problems", kMaxBatch);
    ELIMREC_REQUIRE(d_workspace, "linear_bwd_w: null workspace");
    if (workspace_bytes < elimrec_linear_bwd_w_batched_workspace(descs, n)) {
        set_error("linear_bwd_w: workspace too small (%zu < %zu)", workspace_bytes,
                  elimrec_linear_bwd_w_batched_workspace(descs, n));
        return ELIMREC_E_WORKSPACE;
    }
    BwdBatch batch;
    int blocks = 0, max_out = 0, rc;
    if ((rc = bwd_w_build_batch(descs, n, d_workspace, batch, blocks, max_out))) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (mg) {
        const int merge_blocks = (int)((mg->N + mg->chunk - 1) / mg->chunk);
        static int merge_first = -1;
        if (merge_first < 0) { const char *e = getenv("ELIMREC_MERGE_FIRST"); merge_first = e ? atoi(e) : 1; }
        hipLaunchKernelGGL(linear_bwd_w_merge_kernel, dim3(blocks + merge_blocks), dim3(256), 0, s, batch, *mg, blocks, merge_blocks,
                           merge_first);
    } else {
        hipLaunchKernelGGL(linear_bwd_w_partial_kernel, dim3(blocks), dim3(256), 0, s, batch);
    }
    ELIMREC_LAUNCH_CHECK("linear_bwd_w_partial");
    if (defer_reduce) return 0;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((4 * max_out + 255) / 256, n), dim3(256), 0, s, batch);
    ELIMREC_LAUNCH_CHECK("reduce_slabs");
    return 0;
}

extern "C" int elimrec_linear_bwd_w_reduce(const elimrec_linear_bwd_desc *descs, int n, void *d_workspace, size_t workspace_bytes,
                                           void *stream) {
    ELIMREC_REQUIRE(descs && n >= 1 && n <= kMaxBatch, "linear_bwd_w_reduce: 1..%d problems", kMaxBatch);
    ELIMREC_REQUIRE(d_workspace && workspace_bytes >= bwd_w_batched_bytes(descs, n), "linear_bwd_w_reduce: workspace");
    BwdBatch batch;
    int blocks = 0, max_out = 0, rc;
    if ((rc = bwd_w_build_batch(descs, n, d_workspace, batch, blocks, max_out))) return rc;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((4 * max_out + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, batch);
    ELIMREC_LAUNCH_CHECK("reduce_slabs");
    return 0;
}

extern "C" int elimrec_linear_bwd_w_batched(const elimrec_linear_bwd_desc *descs, int n, void *d_workspace,
                                            size_t workspace_bytes, void *stream) {
    return linear_bwd_w_batched_impl(descs, n, d_workspace, workspace_bytes, nullptr, 0, stream);
}

extern "C" int elimrec_linear_bwd_w_batched_merge(const elimrec_linear_bwd_desc *descs, int n, void *d_workspace,
                                                  size_t workspace_bytes, const float *d_rows, const int32_t *d_keys, int world,
                                                  int64_t R, int64_t U, int64_t I, int ns, int w, int M, float *d_SrcA,
                                                  float *d_SrcB, uint32_t *d_mask, int defer_reduce, void *stream) {
    if (!d_rows) return linear_bwd_w_batched_impl(descs, n, d_workspace, workspace_bytes, nullptr, defer_reduce, stream);
    ELIMREC_REQUIRE(M >= 0, "linear_bwd_w_batched_merge: M >= 0");
    ELIMREC_REQUIRE(d_keys && d_SrcA && d_SrcB && d_mask, "linear_bwd_w_batched_merge: null pointer");
    ELIMREC_REQUIRE(world >= 1 && world <= kSlabMaxRanks && R >= 1 && R < INT32_MAX, "linear_bwd_w_batched_merge: 1..%d ranks", kSlabMaxRanks);
    ELIMREC_REQUIRE(ns >= 1 && w >= 4 && (w & (w - 1)) == 0, "linear_bwd_w_batched_merge: bad slab geometry (ns=%d, w=%d)", ns, w);
    const int64_t N = U + I;
    int sh = 0;
    while ((4 << sh) < w) ++sh;
    MergeArgs mg = {d_rows, d_keys, world, (int)R, U, N, ns * (w / 4), w / 4, sh, merge_rows_chunk(N), M, d_SrcA, d_SrcB, d_mask, 0};
    ELIMREC_REQUIRE(mg.chunk / 32 <= kMergeSeenWords, "linear_bwd_w_batched_merge: %lld rows are more than the fused launch takes "
                    "(call elimrec_slab_merge_rows)", (long long)N);
    return linear_bwd_w_batched_impl(descs, n, d_workspace, workspace_bytes, N > 0 ? &mg : nullptr, defer_reduce, stream);
}

extern "C" int elimrec_linear_bwd_w(const float *d_A, int64_t lda, const float *d_B, int64_t ldb,
                                    const int32_t *d_row_index, const int32_t *d_range, int64_t R, int n1, int n2,
                                    float *d_out, int64_t ldo, float *d_colsum, int accumulate, void *d_workspace,
                                    size_t workspace_bytes, void *stream) {
    elimrec_linear_bwd_desc d = {d_A, lda, d_B, ldb, d_row_index, d_range, R, n1, n2, d_out, ldo, d_colsum, accumulate, nullptr};
    return elimrec_linear_bwd_w_batched(&d, 1, d_workspace, workspace_bytes, stream);
}
