/* libelimrec_hip.so -- C ABI of the MI355X (gfx950) EliMRec hot path.
 *
 * Plain pointers and sizes only: no torch types, no C++ in the signatures. Every pointer
 * argument named d_* is a DEVICE pointer (HBM); `stream` is a hipStream_t passed as void*
 * (NULL = the default stream). All work is enqueued asynchronously on `stream`; nothing here
 * synchronises the host, allocates device memory or frees it (workspace comes from the caller),
 * so every entry point may be captured into a hipGraph.
 *
 * Return value: 0 on success, non-zero on failure (hipError_t value, or ELIMREC_E_*);
 * elimrec_last_error() returns a thread-local message for the most recent failure.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference repository root, Xiaohao-Liu/EliMRec). The reference has no FFI of its own for
 * the model path (it is stock PyTorch); the binding a maintainer adds is the ctypes stub in
 * INTEGRATION.md (elimrec_amd/_lib.py is that stub, as shipped).
 *
 * Layout conventions (all row-major, fp32 unless said otherwise):
 *   N = U + I graph nodes: users first, then items (models/EliMRec.py:239,316).
 *   X  [N x C]   propagated tables, C = M*d: column block m holds table m (0 = id, then v,a,t).
 *   Y  [N x Cy]  head outputs, Cy = (1+S)*d: block 0 = fused embedding (all_users/all_items),
 *                block 1+s = single-modal head s (pre_fusion_{user,item}_{v,a,t}).
 */
#ifndef ELIMREC_HIP_H
#define ELIMREC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ELIMREC_ABI_VERSION 1
#define ELIMREC_E_BADARG 10001
#define ELIMREC_E_UNSUPPORTED 10002
#define ELIMREC_E_WORKSPACE 10003

int elimrec_abi_version(void);
const char *elimrec_last_error(void);

/* ---------------------------------------------------------------- dense projections (K1,K5,K8)
 * C[m, n] = sum_k A[m, k] * W[n, k] + bias[n]      (torch.nn.Linear; bias may be NULL)
 * Replaces: v_dense/a_dense/t_dense (models/EliMRec.py:233-236), embedding_{user,item}_after_GCN
 * (:262-270) and s_dense_{v,a,t} (:146-151). fp32-input MFMA (v_mfma_f32_32x32x2_f32).
 * Leading dimensions in elements. Requires K % 4 == 0, lda % 4 == 0, ldw % 4 == 0 and 16-byte
 * aligned A, W rows. */
int elimrec_linear_fwd(const float *d_A, int64_t lda, const float *d_W, int64_t ldw,
                       const float *d_bias, float *d_C, int64_t ldc,
                       int64_t M, int N, int K, void *stream);

/* Several independent Linears in ONE launch (the three feature projections; the five head Linears):
 * the grid's z dimension indexes the problem. 1 <= n <= 8. */
typedef struct elimrec_linear_desc {
    const float *d_A; int64_t lda;
    const float *d_W; int64_t ldw;
    const float *d_bias;            /* nullable */
    float *d_C; int64_t ldc;
    int64_t M; int32_t N; int32_t K;
    const float *d_rowscale;        /* nullable [M]: the bias term becomes rowscale[m] * bias[n]      */
    const float *d_add; int64_t ldadd;  /* nullable [M x N]: added to the result (C = A.W^T + .. + add) */
    const int32_t *d_row_index;     /* nullable [M]: output row m reads row d_row_index[m] of A, rowscale, add
                                       (the projections evaluated at the batch's rows only; C stays compact) */
    const int32_t *d_row_range;     /* nullable device int32[2] = (begin, end): only output rows [begin, min(end, M))
                                       are produced (slot ranges of elimrec_segment_plan's seg_info)            */
    int32_t act;                    /* epilogue activation: 0 none, 1 relu (util/mlp.py:33-35: act between layers) */
} elimrec_linear_desc;
int elimrec_linear_fwd_batched(const elimrec_linear_desc *descs /* host array */, int n, void *stream);

/* The rows of the node tables one batch touches: rows[3b] = users[b], rows[3b+1] = U + pos[b], rows[3b+2] = U + neg[b]
 * (the gathers all_users[users], all_items[pos], all_items[neg] of models/EliMRec.py:274-289, as node ids). When d_src
 * is not NULL the `cols` leading columns of those rows are copied to d_dst in the same order. */
int elimrec_triplet_rows(const int64_t *d_users, const int64_t *d_pos, const int64_t *d_neg, int64_t B, int64_t U,
                         int32_t *d_rows, const float *d_src, int64_t lds, int cols, float *d_dst, int64_t ldd,
                         void *stream);
/* The same node ids with the range check the reference gets from torch indexing (IndexError / device assert on
 * all_users[users], all_items[pos], models/EliMRec.py:277-281): an index outside [0, U) / [0, I) sets bit 0 / 1 / 2
 * (user / positive / negative) of *d_err and is replaced by 0, so nothing downstream reads or writes out of bounds;
 * the host tests the word when it next synchronises (EliMRec.check_indices()). */
int elimrec_triplet_rows_checked(const int64_t *d_users, const int64_t *d_pos, const int64_t *d_neg, int64_t B,
                                 int64_t U, int64_t I, int32_t *d_rows, int32_t *d_err, void *stream);
/* The whole index front end of a training step in ONE launch (for 3B <= 8192 slots; more fall back to separate
 * launches): node ids of the triplet slots (range-checked as above) -> d_keys, then elimrec_segment_plan of those
 * keys (split_key = U, key_space = U + I), then the unused tail of d_active_rows set to pad_key + slot (distinct
 * negative keys: what the ranks of a column-sharded job exchange and merge). */
int elimrec_batch_plan(const int64_t *d_users, const int64_t *d_pos, const int64_t *d_neg, int64_t B, int64_t U,
                       int64_t I, int32_t *d_keys, int32_t *d_active_rows, int32_t *d_seg_info,
                       int32_t *d_slot_seg, uint32_t *d_key_bitmap, int32_t pad_key, int32_t *d_err,
                       void *d_workspace, size_t workspace_bytes, void *stream);
/* dst[r, 0:cols] = src[rows[r], 0:cols] for r < min(*d_count, n)  (d_count nullable; cols % 4 == 0). */
int elimrec_gather_rows(const float *d_src, int64_t lds, const int32_t *d_rows, const int32_t *d_count, int64_t n,
                        int cols, float *d_dst, int64_t ldd, void *stream);

/* out[i, j] (+)= sum_{r in rows} A[r, i] * B[row_index ? row_index[r] : r, j]
 * the weight-gradient contraction of a Linear layer (AddmmBackward of the calls above) with a
 * deterministic two-stage reduction: fixed row chunks -> partial slabs in workspace -> summed in
 * chunk order. Rows r run over [range[0], range[1]) if d_range != NULL (device int32[2]) else
 * [0, R). d_colsum (nullable, [n1]) additionally receives sum_r A[r, i] (the bias gradient).
 * workspace: elimrec_linear_bwd_w_workspace(R, n1, n2) bytes. */
size_t elimrec_linear_bwd_w_workspace(int64_t R, int n1, int n2);
int elimrec_linear_bwd_w(const float *d_A, int64_t lda, const float *d_B, int64_t ldb,
                         const int32_t *d_row_index, const int32_t *d_range, int64_t R,
                         int n1, int n2, float *d_out, int64_t ldo, float *d_colsum,
                         int accumulate, void *d_workspace, size_t workspace_bytes, void *stream);

/* Batched form: n independent weight-gradient contractions in one partial launch + one reduce
 * launch. 1 <= n <= 8. workspace: elimrec_linear_bwd_w_batched_workspace(descs, n) bytes. */
typedef struct elimrec_linear_bwd_desc {
    const float *d_A; int64_t lda;
    const float *d_B; int64_t ldb;
    const int32_t *d_row_index;     /* nullable */
    const int32_t *d_range;         /* nullable device int32[2] */
    int64_t R; int32_t n1; int32_t n2;
    float *d_out; int64_t ldo;
    float *d_colsum;                /* nullable */
    int32_t accumulate;
    const float *d_colsum_weight;   /* nullable: colsum becomes sum_r w[row(r)] * A[r, i], row(r) as for B (the bias
                                       gradient of a folded projection, w = c) */
} elimrec_linear_bwd_desc;
size_t elimrec_linear_bwd_w_batched_workspace(const elimrec_linear_bwd_desc *descs, int n);
int elimrec_linear_bwd_w_batched(const elimrec_linear_bwd_desc *descs /* host array */, int n,
                                 void *d_workspace, size_t workspace_bytes, void *stream);
/* The same, with elimrec_slab_merge_rows (arguments as there, declared below) as extra workgroups of the partial
 * launch: both consume the head backward's dOut rows and nothing of each other (IndexBackward into the adjoint sources
 * next to AddmmBackward's dW -- /root/reference/models/EliMRec.py:262-270 under autograd), so the latency-bound merge
 * costs no launch of its own. Same bits as the two calls. (U + I) / 1024 <= 131072 rows per workgroup. */
int elimrec_linear_bwd_w_batched_merge(const elimrec_linear_bwd_desc *descs /* host array */, int n, void *d_workspace,
                                       size_t workspace_bytes, const float *d_rows, const int32_t *d_keys, int world,
                                       int64_t R, int64_t U, int64_t I, int ns, int w, int M, float *d_SrcA,
                                       float *d_SrcB, uint32_t *d_mask, int defer_reduce, void *stream);
/* defer_reduce != 0: the launch above ends with the partial slabs in the workspace; the caller runs the fixed-order
 * reduce later, before the gradients are read -- by this call, or as extra workgroups of the adjoint's first hop
 * (elimrec_slab_hop_bwd_w below). d_rows == NULL: no merge. Same descs / workspace as the partial launch. */
int elimrec_linear_bwd_w_reduce(const elimrec_linear_bwd_desc *descs /* host array */, int n, void *d_workspace,
                                size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------- layer-0 table assembly (K2)
 * X0[u, m*d + j] = user_emb[u, j] for every table m;  X0[U+i, j] = item_emb[i, j].
 * Column blocks 1..M-1 of the item rows are written by elimrec_linear_fwd (ldc = C).
 * Replaces torch.cat([u_emb, i_emb]) x M (models/EliMRec.py:239,250-256). */
int elimrec_assemble_x0(const float *d_user_emb, const float *d_item_emb, float *d_X0,
                        int64_t U, int64_t I, int d, int M, void *stream);

/* ---------------------------------------------------------------- LightGCN propagation (K3,K4)
 * One hop over a CSR matrix (int32 rowptr[n_rows+1], int32 col, fp32 val) for all C columns:
 *   r = sum_j val[j] * Xin[col[j], :]
 *   if Xout   : Xout[row]   = r
 *   if AccOut : AccOut[row] = (AccIn[row] + r) * scale        (AccIn may alias AccOut)
 * Replaces torch.sparse.mm (models/EliMRec.py:244) fused with the stack+mean of :246-247.
 * Requires C % 4 == 0.
 *
 * Load balance for the power-law head (a popular item has ~10^4 neighbours, the median node ~10):
 * an optional row-split plan, built once per matrix by the caller, cuts every row with more than
 * `long_threshold` non-zeros into segments of at most that many; segments are summed by separate
 * waves into d_partials [n_seg x C] and combined per row in segment order (deterministic). */
typedef struct elimrec_csr_split {
    int32_t long_threshold;         /* rows with nnz > threshold are split                      */
    int32_t n_long;                 /* number of split rows (0 => plan unused)                  */
    int32_t n_seg;                  /* total segments over all split rows                       */
    const int32_t *d_long_rows;     /* [n_long] row ids                                         */
    const int32_t *d_long_seg_ptr;  /* [n_long+1] segment range of each split row               */
    const int32_t *d_seg_bounds;    /* [n_seg][2] (begin,end) positions into col/val            */
    float *d_partials;              /* [n_seg x 2C] scratch (C-column region, then d-column region) */
    const int32_t *d_seg_row;       /* [n_seg] index into d_long_rows of each segment's row (nullable) */
    const int32_t *d_row_order;     /* [n_rows] (nullable): order in which rows are handed to waves -- sorted
                                       by length so the 64/LPR rows sharing a wave finish together; a pure
                                       scheduling hint (used by the narrow-row kernels), results unchanged */
    int32_t *d_tickets;             /* [2*n_long] zero-initialised arrival counters (nullable): when
                                       given, the wave that finishes a split row's LAST segment combines
                                       the row in segment order inside the same launch (agent-scope
                                       release/acquire); otherwise a separate fix-up launch does. Counters
                                       return to zero after every launch.                          */
    const int32_t *d_row_items;     /* [n_row_items][3] (nullable): (row, begin, end) of every row that is NOT split,
                                       in processing order (d_row_order's). With it the narrow-row kernel runs as a
                                       persistent stream: a wave fetches the triple of item k+2 and the first
                                       neighbour indices of item k+1 while it gathers item k, so a row costs one
                                       dependent memory round trip instead of four. Results unchanged.             */
    int32_t n_row_items;
} elimrec_csr_split;

int elimrec_spmm_hop(const int32_t *d_rowptr, const int32_t *d_col, const float *d_val,
                     int64_t n_rows, int C, const elimrec_csr_split *split /* nullable */,
                     const float *d_Xin, float *d_Xout,
                     const float *d_AccIn, float *d_AccOut, float scale, void *stream);

typedef struct elimrec_csr {
    int64_t n_rows;
    const int32_t *d_rowptr;        /* [n_rows+1] */
    const int32_t *d_col;
    const float *d_val;
    elimrec_csr_split split;        /* n_long == 0 => no row-split plan */
} elimrec_csr;

/* L hops + mean of the L+1 layer outputs: Out = 1/(L+1) * sum_k A^k X0   (compute_graph,
 * models/EliMRec.py:238-248, for all M tables at once). d_tmp0/d_tmp1: two [n_rows x C]
 * scratch tables (unused when L <= 1 / L <= 2). X0 is left intact. Out must not alias X0. */
int elimrec_propagate(const int32_t *d_rowptr, const int32_t *d_col, const float *d_val,
                      int64_t n_rows, int C, const elimrec_csr_split *split /* nullable */, int L,
                      const float *d_X0, float *d_tmp0, float *d_tmp1, float *d_Out, void *stream);

/* Bipartite form of the same propagation, for adjacencies without a diagonal ('pre', 'plain',
 * 'gcmc': A = [[0, P], [Q, 0]], P = users<-items [U x I], Q = items<-users [I x U], column ids
 * local to the source side). The layer-0 table is [E_u ; XI] with the user rows shared by all M
 * tables (models/EliMRec.py:250-256), so layer k splits into a C-column chain alive on one side and
 * a d-column chain (identical for every table) alive on the other: each hop costs nnz/2 edges at C
 * columns + nnz/2 at d columns instead of nnz at C. Results are identical to elimrec_propagate:
 *   Out[N x C] = 1/(L+1) * sum_k A^k [bcast(E_u) ; XI].          Requires L >= 1.
 * workspace: elimrec_bipartite_workspace(U, I, d, M) bytes (shared by forward and backward). */
size_t elimrec_bipartite_workspace(int64_t U, int64_t I, int d, int M);
int elimrec_propagate_bipartite(const elimrec_csr *P, const elimrec_csr *Q, int64_t U, int64_t I,
                                int d, int M, int L, const float *d_user_emb /* [U x d] */,
                                const float *d_XI /* [I x C] */, float *d_Out /* [N x C] */,
                                float *d_narrow_out /* nullable [N x d]: 1/(L+1) * sum_k A^k [E_u ; 0], the part of
                                                       Out that every table shares */,
                                void *d_workspace, size_t workspace_bytes, void *stream);

/* The C-column and d-column chains of the bipartite propagation are independent for most of a call;
 * optionally (elimrec_set_concurrency(1); default off -- the step is bandwidth
 * bound, measured gain 0.6 %) the d-column chain runs on an internal side stream forked from / joined to
 * `stream` with events (capturable). The row-split scratch d_partials must hold [n_seg x 2C] floats. */
int elimrec_ticket_fixup(void);               /* 1 (default): in-launch combine when d_tickets is given */
void elimrec_set_ticket_fixup(int on);
int elimrec_concurrency(void);
void elimrec_set_concurrency(int on);

/* Its adjoint (SparseAddmmBackward x L, StackBackward/MeanBackward, CatBackward of the reference's
 * autograd): from G = dLoss/dOut [N x C], non-zero only on the rows listed in d_active_rows
 * (first d_seg_info[0] entries; other rows of G are never read, so G need not be zero-filled) and
 * H[r, :] = sum_m G[r, m*d:(m+1)*d] on those rows, computes
 *   gXI [I x C] = dLoss/dXI   and   gE_u [U x d] = dLoss/dE_u.
 * PT = P^T [I x U], QT = Q^T [U x I] (for a symmetric adjacency PT == Q and QT == P). */
int elimrec_propagate_bipartite_bwd(const elimrec_csr *PT, const elimrec_csr *QT, int64_t U, int64_t I,
                                    int d, int M, int L, const float *d_G, const float *d_H,
                                    const int32_t *d_active_rows, const int32_t *d_seg_info,
                                    int64_t n_max, float *d_gXI, float *d_gEu, void *d_workspace,
                                    size_t workspace_bytes, void *stream);

/* Folded propagation (the default path): when the constant feature tables are folded into GEMM operands
 *  only the d-column table X0 = [E_u ; E_i] goes through the graph, one launch per hop over
 * the FULL adjacency A [N x N] (no diagonal blocks):
 *   Out0[row, 0:d] (row stride ldo) = 1/(L+1) sum_k A^k X0
 *   narrow[N x d] = the part of it that comes from E_u alone (= what every feature table shares)
 * and its adjoint from the slot-major dOut rows [n x M*d] of the active nodes:
 *   grad[N x d] = [dLoss/dE_u ; dLoss/dE_i]   (SrcA/SrcB: two [N x d] scratch tables, written on active rows)
 * AT = A^T (A itself when symmetric). workspace: elimrec_folded_workspace(N, d) bytes for either call. */
size_t elimrec_folded_workspace(int64_t N, int d);
int elimrec_propagate_folded(const elimrec_csr *A, int64_t U, int64_t I, int d, int L, const float *d_X0,
                             float *d_Out0, int64_t ldo, float *d_narrow, void *d_workspace,
                             size_t workspace_bytes, void *stream);
int elimrec_propagate_folded_bwd(const elimrec_csr *AT, int64_t U, int64_t I, int d, int M, int L,
                                 const float *d_dOutR, const int32_t *d_active_rows, const int32_t *d_seg_info,
                                 int64_t n_max, float *d_SrcA, float *d_SrcB, float *d_grad,
                                 const uint32_t *d_active_mask /* nullable: bitmap of the active rows (the key bitmap of
                                 elimrec_segment_plan); built internally when NULL; required when d_dOutR is NULL = sources
                                 prefilled by the caller) */,
                                 void *d_workspace, size_t workspace_bytes, void *stream);

/* [H | G] of every active row -- H = sum of the M column blocks of its dOut row (block order), G = block 0: all the adjoint
 * propagation needs of a dOut row -- cut into the column slices a column-sharded job sends to its `world` peers (the
 * "sparse-grad" exchange, elimrec_amd/shard.py; the receiving side sums rows of one node in rank order with
 * elimrec_slab_merge_rows): d_out [world x n_max x 2*dl], dl = d / world: slice w of row s =
 * [H[s][w*dl:(w+1)*dl] | G[s][w*dl:(w+1)*dl]]. */
int elimrec_source_rows_split(const float *d_dOutR, const int32_t *d_count, int64_t n_max, int d, int M, int world,
                              float *d_out, void *stream);
/* One block SpMM with the fused epilogue on a W-column window of wider tables (row stride ld):
 *   r = A . Xin[:, 0:W];  if Xout: Xout = r;  if AccOut: AccOut = (r + Add1) * scale.
 * The building block of the bipartite propagation, exposed for callers that tile columns themselves. */
int elimrec_block_spmm(const elimrec_csr *A, int W, int64_t ld, const float *d_Xin, float *d_Xout,
                       const float *d_Add1, float *d_AccOut, float scale, void *stream);

/* H[r, j] = sum_m G[r, m*d + j] for the active rows r (slots < d_seg_info[0]). */
int elimrec_blocksum_rows(const float *d_G, const int32_t *d_active_rows, const int32_t *d_seg_info,
                          int64_t n_max, int d, int M, int slot_major /* G rows indexed by slot, not node */,
                          float *d_H, void *stream);

/* dst[r, 0:n_cols] = src[r, 0:n_cols] with independent leading dimensions (column-block copies:
 * E_item into block 0 of XI; block 0 of gXI into the embedding_item gradient). n_cols % 4 == 0. */
int elimrec_copy_cols(const float *d_src, int64_t ld_src, float *d_dst, int64_t ld_dst, int64_t n_rows,
                      int n_cols, void *stream);

/* ---------------------------------------------------------------- cosine-BPR head (K7,K9,K10)
 * For triplet b and head block k (weight w[k]; w[k] == 0 skips the block):
 *   a = Y[users[b]], p = Y[U+pos[b]], n = Y[U+neg[b]]  (block k of each row)
 *   x = cos(a,n) - cos(a,p);  loss_b += w[k] * softplus(x) / B
 * d_loss_rows[b] = loss_b. If d_grad_rows != NULL it receives d(sum_b loss_b)/dY for the three
 * gathered rows of every triplet as [3B x Cy] rows (order u,p,n per triplet) and d_keys[3B]
 * the node ids (u, U+p, U+n) they belong to. Replaces getEmbedding's gathers + original_bpr_loss
 * x (1+|modality|) + their autograd (models/EliMRec.py:129-142,277-287,291-297).
 * F.normalize eps 1e-12, softplus threshold 20 as in torch. */
int elimrec_bpr_head(const float *d_Y, int64_t ldy, int64_t U, int64_t I,
                     const int64_t *d_users, const int64_t *d_pos, const int64_t *d_neg, int B,
                     int d, int n_blocks, const float *block_weights /* host, n_blocks */,
                     float *d_loss_rows, float *d_grad_rows, int32_t *d_keys, void *stream);

/* The same loss over a COMPACT table: slot 3b+j of triplet b reads row d_slot_rows[3b+j] of d_Y (the rows of the
 * batch's active nodes, elimrec_segment_plan's slot -> segment map). No keys are written. */
int elimrec_bpr_head_rows(const float *d_Y, int64_t ldy, const int32_t *d_slot_rows, int B, int d, int n_blocks,
                          const float *block_weights /* host, n_blocks */, float *d_loss_rows, float *d_grad_rows,
                          void *stream);
/* elimrec_bpr_head_rows + elimrec_sum of its loss rows in ONE launch: the workgroup that finishes last adds the B loss
 * rows in elimrec_sum's order (same bits) into *d_loss. d_ticket: one int32, zero before the first call (the kernel
 * leaves it zero). */
int elimrec_bpr_head_rows_sum(const float *d_Y, int64_t ldy, const int32_t *d_slot_rows, int B, int d,
                              int n_blocks, const float *block_weights, float *d_loss_rows, float *d_grad_rows,
                              float *d_loss, int32_t *d_ticket, void *stream);

/* ... and the batch loss PUBLISHED to the host from that launch. /root/reference/main.py:102 reads `loss.cpu().item()` after
 * every step; a read of the device tensor waits for the whole step (adjoint hops, Adam), so the host could not enqueue step t + 1
 * under step t. Here the launch that sums the loss (about 120 us into a 280 us step) also stores (sequence number, value) as one
 * 8-byte system-scope word into coherent host memory, and the caller's `.item()` waits on that word -- not on the stream.
 * pub: elimrec_loss_pub_create(n_slots) (a ring of n_slots host words + a device-side launch counter). Every
 * elimrec_bpr_head_rows_sum_pub enqueue publishes under the next sequence number (elimrec_loss_pub_issued right after the enqueue);
 * elimrec_loss_pub_wait(pub, seq, timeout_s, &value) spins until launch `seq` has published: 0 = value is its loss (the bits of
 * *d_loss); ELIMREC_E_UNSUPPORTED = the ring has wrapped past it (read the device tensor); ELIMREC_E_WORKSPACE = timed out. */
int elimrec_loss_pub_create(int n_slots, void **out_pub);
int elimrec_loss_pub_destroy(void *pub);
uint32_t elimrec_loss_pub_issued(void *pub);
int elimrec_loss_pub_wait(void *pub, uint32_t seq, double timeout_s, float *value);
int elimrec_bpr_head_rows_sum_pub(const float *d_Y, int64_t ldy, const int32_t *d_slot_rows, int B, int d,
                                  int n_blocks, const float *block_weights, float *d_loss_rows, float *d_grad_rows,
                                  float *d_loss, int32_t *d_ticket, void *pub, void *stream);

/* out[0] = sum_i x[i] in a fixed order (single workgroup, deterministic). */
int elimrec_sum(const float *d_x, int64_t n, float *d_out, void *stream);

/* ---------------------------------------------------------------- deterministic row scatter-add
 * IndexBackward/index_put(accumulate) replacement: sorts (key, source row) pairs
 * (stable radix sort), sums rows with equal keys in ascending source order.
 * Outputs: d_active_rows[n] sorted unique keys (first n_active valid), d_reduced [n x ld] the
 * summed rows, d_seg_info int32[8] = {n_active, n_lo = #active keys < split_key,
 * (0, n_lo), (n_lo, n_active), (0, n_active)}: the three (begin,end) pairs are slot ranges in
 * the form elimrec_linear_bwd_w's d_range takes (users / items / all, with split_key = U).
 * workspace: elimrec_segment_reduce_workspace(n) bytes. */
size_t elimrec_segment_reduce_workspace(int64_t n);
/* The two halves of elimrec_segment_reduce_rows. The PLAN depends on the keys only (the batch's node ids are known
 * before the forward pass): sorted unique keys -> d_active_rows, d_seg_info as above, d_slot_seg[n] = the segment
 * (index into d_active_rows) of every slot, member lists inside the workspace. key_space > 0 promises
 * 0 <= key < key_space and selects the one-workgroup bitmap planner for n <= 8192 when the bitmap fits LDS (same
 * output as the radix-sort path, which is used otherwise). APPLY sums rows per segment in ascending slot order, times *d_scale. */
size_t elimrec_segment_plan_workspace(int64_t n);
int elimrec_segment_plan(const int32_t *d_keys, int64_t n, int32_t split_key, int64_t key_space,
                         int32_t *d_active_rows, int32_t *d_seg_info, int32_t *d_slot_seg,
                         uint32_t *d_key_bitmap /* nullable [(key_space+31)/32]: bit k set <=> k is an active key */,
                         void *d_workspace, size_t workspace_bytes, void *stream);
int elimrec_segment_apply(const float *d_rows, int64_t n, int ld, const int32_t *d_seg_info, const float *d_scale,
                          float *d_reduced, const void *d_workspace, size_t workspace_bytes, void *stream);
int elimrec_segment_reduce_rows(const float *d_rows, const int32_t *d_keys, int64_t n, int ld,
                                int32_t split_key, int32_t *d_active_rows, float *d_reduced,
                                const float *d_scale /* nullable device fp32[1]: multiplies every sum */,
                                int32_t *d_seg_info, void *d_workspace, size_t workspace_bytes,
                                void *stream);

/* ---------------------------------------------------------------- head backward wrt its input
 * For active row s < n_active (node r = active_rows[s]):
 *   G0[r, :] = dY[s, 0:d] . Wf(r)  +  sum_h dY[s, (1+h)*d : (2+h)*d] . Ws_h  placed in block mblock[h]
 * with Wf = W_user [d x C] for r < U else W_item. G0 must be zero-filled by the caller; rows
 * not in the active list stay zero. `gscale` multiplies everything (upstream d loss).
 * AddmmBackward (dX) of embedding_*_after_GCN and s_dense_* restricted to the rows whose
 * gradient is non-zero. head_mblock: host int[S], table index (1..M-1) feeding head h. */
int elimrec_head_bwd_input(const float *d_dY, int64_t lddy, const int32_t *d_active_rows,
                           const int32_t *d_seg_info, int64_t n_max, int64_t U, int d, int C,
                           int S, const int *head_mblock, const float *d_W_user,
                           const float *d_W_item, const float *const *d_W_heads /* host array of S device ptrs */,
                           float gscale, float *d_G0 /* nullable: scattered rows G0[node, 0:scatter_cols], stride ldg */,
                           int64_t ldg, int scatter_cols,
                           float *d_compact /* nullable [n_max x C]: the same rows in slot order */, void *stream);

/* elimrec_segment_apply followed by elimrec_head_bwd_input (compact output only) as ONE launch: the rows of dY are
 * summed from their member gradient rows while they are staged for the contraction; d_reduced still receives dY
 * (elimrec_linear_bwd_w reads it). Same results as the two calls, bit for bit. ld = (1+S)*d. */
int elimrec_segment_apply_head_bwd(const float *d_rows, int64_t n, int ld, const int32_t *d_active_rows,
                                   const int32_t *d_seg_info, const float *d_scale, float *d_reduced,
                                   const void *d_plan_workspace, size_t plan_workspace_bytes, int64_t U, int d, int C,
                                   int S, const int *head_mblock, const float *d_W_user, const float *d_W_item,
                                   const float *const *d_W_heads, float *d_compact, void *stream);
/* The same with the weight operands taken from the packed copy elimrec_head_fwd_fused (phase 0/1, 16-row form) leaves
 * behind: d_pack_bwd = d_pack + elimrec_head_pack_bwd_offset(n_mod, D) floats; recdim 64, head h = feature table h. */
int elimrec_segment_apply_head_bwd_packed(const float *d_rows, int64_t n, int ld, const int32_t *d_active_rows,
                                          const int32_t *d_seg_info, const float *d_scale, float *d_reduced,
                                          const void *d_plan_workspace, size_t plan_workspace_bytes, int64_t U,
                                          int d, int C, int S, const int *head_mblock, const float *d_W_user,
                                          const float *d_W_item, const float *const *d_W_heads,
                                          float *d_compact, const float *d_pack_bwd, void *stream);
/* The packed form for ONE rank that owns every table column (recdim 64): the kernel also writes what
 * elimrec_slab_merge_rows(world = 1, M = C / d) would make of d_compact -- the slab-major adjoint sources [ns x N x w]
 * (H = sum of a row's column blocks in block order, G = block 0; users: H -> SrcA, G -> SrcB, items the other way round)
 * at the active rows. Every active row is listed once, so there is nothing to accumulate; the row bitmap of the
 * sources is the planner's key bitmap (elimrec_batch_plan). IndexBackward's index_put into the adjoint sources
 * (/root/reference/models/EliMRec.py:239-256 under autograd) without a launch of its own. */
int elimrec_segment_apply_head_bwd_sources(const float *d_rows, int64_t n, int ld, const int32_t *d_active_rows,
                                           const int32_t *d_seg_info, const float *d_scale, float *d_reduced,
                                           const void *d_plan_workspace, size_t plan_workspace_bytes, int64_t U,
                                           int d, int C, int S, const int *head_mblock, const float *d_W_user,
                                           const float *d_W_item, const float *const *d_W_heads,
                                           float *d_compact, const float *d_pack_bwd, int64_t N, int ns, int w,
                                           float *d_SrcA, float *d_SrcB, void *stream);
/* ... and with peers: the kernel also writes what elimrec_source_rows_split(d_compact) would -- the [H | G] rows cut into
 * the `world` column slices of a column-sharded job, d_out [world x n_max x 2*(d/world)] -- ready for the all-to-all. */
int elimrec_segment_apply_head_bwd_split(const float *d_rows, int64_t n, int ld, const int32_t *d_active_rows,
                                         const int32_t *d_seg_info, const float *d_scale, float *d_reduced,
                                         const void *d_plan_workspace, size_t plan_workspace_bytes, int64_t U,
                                         int d, int C, int S, const int *head_mblock, const float *d_W_user,
                                         const float *d_W_item, const float *const *d_W_heads,
                                         float *d_compact, const float *d_pack_bwd, int64_t n_max, int world,
                                         float *d_out, void *stream);

/* ---------------------------------------------------------------- embedding gradients (K2 bwd)
 * dE_user[u, j] = sum_m G[u, m*d + j];  dE_item[i, j] = G[U+i, j]   (CatBackward of :239). */
int elimrec_embed_grad(const float *d_G, int64_t U, int64_t I, int d, int M,
                       float *d_grad_user, float *d_grad_item, void *stream);

/* ---------------------------------------------------------------- optimiser (K11)
 * torch.optim.Adam single-tensor step with coupled L2 (main.py:49,101):
 *   g += wd*p; m = m + (1-b1)(g-m); v = b2*v + (1-b2) g*g;
 *   p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * over n contiguous elements. `step` is the 1-based step count t. */
int elimrec_adam_step(float *d_p, const float *d_g, float *d_m, float *d_v, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay,
                      int64_t step, void *stream);

/* ---------------------------------------------------------------- counterfactual scoring (K12-K16)
 * scores[b, i] for users rows `d_users` (node ids < U) against every item:
 *   ui = sigmoid(<Yf[u], Yf[U+i]>)
 *   z_h = <Ys_h[u]/|.|, Ys_h[U+i]/|.|>                     (general_cm_fusion, normalize=True)
 *   fusion_mode 0 (rubi): f(x) = x * prod_{h in mask} sigmoid(z_h)
 *               1 (hm)  : t = sigmoid(x) * prod_h sigmoid(z_h); f = log(t+1e-12) - log1p(t)
 *               2 (sum) : f = log(sigmoid(x + sum_h z_h) + 1e-12)
 *   predict_type 0 (normal): sigmoid(ui) ; 1 (TE): sigmoid(f(ui)) ;
 *                2 (TIE): sigmoid(f(ui) - f(mean_i ui))
 * Replaces EliMRec.predict + general_cm_fusion (models/EliMRec.py:96-113,155-212).
 * d_train_ptr/d_train_items (nullable): CSR over the B rows of items to overwrite with -inf
 * (cpp/uni_evaluator.py:149-154). d_scores [B x I] may be NULL when only top-K is wanted.
 * d_topk_idx/d_topk_val [B x K] (nullable): per-row top-K by (score desc, index asc).
 * workspace: elimrec_score_workspace2(B, U, I, S, K) bytes. */
size_t elimrec_score_workspace(int B, int64_t I, int K);              /* without the norm table   */
size_t elimrec_score_workspace2(int B, int64_t U, int64_t I, int S, int K);
/* Workspace when ONLY top-K is requested (d_scores == NULL, K <= 256, recdim 32 / 64 / 128): no [B x I] score block -- the
 * catalogue goes through the scorer 16384 items at a time, every chunk leaves its K best (id, score) pairs per user and a
 * last launch merges them (same list as the whole-catalogue selection). */
size_t elimrec_score_workspace_topk(int B, int64_t U, int64_t I, int S, int K);   /* the chunked layout itself */
/* What elimrec_score_topk needs for THIS call shape (recdim d, K, score matrix asked for or not): the chunked layout exactly
 * when the call takes the chunked form, else the full [B x I] layout -- one predicate inside the library decides both, so
 * any recdim and any K the reference accepts (models/EliMRec.py:96-113) get a workspace that fits. */
size_t elimrec_score_workspace_for(int B, int64_t U, int64_t I, int S, int K, int d, int want_scores);
/* Item-sharded evaluation (last row; reference: models/EliMRec.py:96-113 scores the whole catalogue, the
 * mean of :107 runs over all items; evaluator/backend/cpp/uni_evaluator.py:131-185 ranks it): this rank holds the cached rows
 * of items [id_offset, id_offset + I) of I_total -- d_Y = [all U user rows ; MY I item rows], train_items already restricted
 * to my range and shifted to local ids. phase 1 (TIE only, otherwise a no-op): d_row_sum[b] = sum over MY items of
 * sigmoid(u.i), summed in fixed order; the caller adds the shards' sums (all_reduce over xGMI). phase 2: scores / top-K of
 * my items with the row mean d_row_sum[b] / I_total; top-K ids are catalogue ids (local id + id_offset). The K best of the
 * whole catalogue are elimrec_topk_merge of the shards' lists. Workspace as for elimrec_score_topk with I = my item count. */
int elimrec_score_topk_shard(const float *d_Y, int64_t ldy, int64_t U, int64_t I, const int64_t *d_users, int B, int recdim,
                             int S, uint32_t head_mask, int fusion_mode, int predict_type, const float *d_sqnorm,
                             const int64_t *d_train_ptr, const int32_t *d_train_items, float *d_scores, int64_t lds, int K,
                             int32_t *d_topk_idx, float *d_topk_val, void *d_workspace, size_t workspace_bytes, int phase,
                             float *d_row_sum, int64_t I_total, int64_t id_offset, void *stream);
/* d_cand_val / d_cand_idx [B x n_cand] (idx < 0: no candidate) -> per row the K best by (score desc, item id asc). */
int elimrec_topk_merge(const float *d_cand_val, const int32_t *d_cand_idx, int B, int n_cand, int K, int32_t *d_topk_idx,
                       float *d_topk_val, void *stream);
/* Evaluation math of the scorer: 0 = EXACT (IEEE division, libm expf), 1 = FAST (default: sigmoids through v_exp_f32 with
 * a two-float argument product and v_rcp_f32 + one Newton step, reciprocal norms refined the same way -- every factor
 * within ~2 ulp of the EXACT form, scores within 1.2e-7 absolute, a validation pass 19 % shorter). Both are within 1e-6 of
 * the reference's scores (/root/reference/models/EliMRec.py:155-212). Also env ELIMREC_EVAL_MATH=exact, read once. */
void elimrec_score_set_math(int mode);
int elimrec_score_get_math(void);
/* FAST math, chunked top-K (no score matrix), recdim 32 / 64: both scorer passes run on the bf16 matrix cores from EXACT
 * three-piece splits of the fp32 operands (x = x1 + x2 + x3, bf16 each; the six piece products above 2^-24 relative,
 * accumulated in fp32): the same scores to fp32 round-off (within 2.4e-7 of the EXACT mode, tests) at 2.7x less matrix-core
 * time. On by default (this switch turns it off); needs the workspace of
 * elimrec_score_workspace_for (room for one chunk's pieces), otherwise the fp32 MFMA form runs. */
void elimrec_score_set_bf16x3(int on);
int elimrec_score_get_bf16x3(void);

/* Range invariant of the scorer (recdim 32 / 64 / 128 forms): every score of a valid (user, item) pair is the last sigmoid of a
 * bounded argument -- sigma([0, 1]) for predict type normal and rubi TE, sigma([-1, 1]) for rubi TIE (models/EliMRec.py:96-113,
 * 171-188), [0, 1] for the logarithmic fusions -- and a TIE row mean lies in (0, 1). Every elimrec_score_topk* call checks, in a
 * last small launch, every score it RETURNS in a K-list (a wrongly high score necessarily enters its user's list) and every row
 * mean it used; an offending user row adds 1 to a device counter. *h_count = the counter, read behind everything enqueued on
 * `stream` (synchronises with it); reset != 0 clears it behind the read. The evaluator reads it after every pass. (The same
 * comparison inside the scorers' epilogues cost 1.5 - 6 % of a validation pass and is not taken.) */
int elimrec_score_range_violations(int64_t *h_count, int reset, void *stream);
/* ... the same check over lists the caller holds: d_topk_val / d_topk_idx [B x K] (ids < 0 and -inf scores are fillers, not
 * checked), d_row_mean [B] (nullable). */
int elimrec_score_range_check(const float *d_topk_val, const int32_t *d_topk_idx, int B, int K, int predict_type, int fusion_mode,
                              const float *d_row_mean, void *stream);

/* d_sqnorm (nullable): [N x (1+S)] squared norms of every head block of every row of Y, from
 * elimrec_row_sqnorms; pass it when several user blocks are scored against the same tables (an
 * evaluation pass), otherwise it is recomputed inside every call. */
int elimrec_row_sqnorms(const float *d_Y, int64_t ldy, int64_t n_rows, int d, int n_blocks, float *d_out,
                        void *stream);
int elimrec_score_topk(const float *d_Y, int64_t ldy, int64_t U, int64_t I, const int64_t *d_users,
                       int B, int d, int S, uint32_t head_mask, int fusion_mode, int predict_type,
                       const float *d_sqnorm, const int64_t *d_train_ptr, const int32_t *d_train_items,
                       float *d_scores, int64_t lds, int K, int32_t *d_topk_idx, float *d_topk_val,
                       void *d_workspace, size_t workspace_bytes, void *stream);

/* The reference's order among EQUAL scores. /root/reference/evaluator/backend/cpp/include/evaluate.h:26-33 ranks a user's masked
 * score row with std::partial_sort_copy over the item ids under comp(x1, x2) = ratings[x1] > ratings[x2]; among equal scores its
 * lists are the heap order of the C++ library's algorithm (first K ids -> make_heap; every later id that beats the heap's top ->
 * __adjust_heap; sort_heap), not an order of the ids. Three entry points give those lists bit for bit:
 *  - elimrec_score_topk_ordered: elimrec_score_topk with tie_order 1 -- ON THE DEVICE, inside the scoring call: one wave per user
 *    row replays the algorithm operation for operation (csrc/eval.hip ref_order_kernel), scanning 64 items (or the maxima of 64
 *    sixteen-item tiles) per step against the heap's top; a catalogue scored chunk by chunk carries the heap between the chunks.
 *    No score row leaves the device, no host synchronisation; K <= 1024. tie_order 0 = elimrec_score_topk.
 *  - elimrec_topk_reference_order_device: the same kernel over rows of masked scores already on the device, [n_rows x ld] ->
 *    d_topk_idx / d_topk_val (nullable) [n_rows x K].
 *  - elimrec_topk_reference_order: a HOST function over host rows running std::partial_sort_copy of the C++ library this package
 *    is built with (what the device kernel is tested against, beside the reference's own compiled evaluate.h). */
int elimrec_score_topk_ordered(const float *d_Y, int64_t ldy, int64_t U, int64_t I, const int64_t *d_users,
                               int B, int d, int S, uint32_t head_mask, int fusion_mode, int predict_type,
                               const float *d_sqnorm, const int64_t *d_train_ptr, const int32_t *d_train_items,
                               float *d_scores, int64_t lds, int K, int32_t *d_topk_idx, float *d_topk_val,
                               void *d_workspace, size_t workspace_bytes, int tie_order, void *stream);
int elimrec_topk_reference_order_device(const float *d_scores, int64_t n_rows, int64_t I, int64_t ld, int K,
                                        int32_t *d_topk_idx, float *d_topk_val, void *stream);
int elimrec_topk_reference_order(const float *h_scores, int64_t n_rows, int64_t I, int64_t ld, int K, int32_t *h_topk);

/* Precision/Recall/MAP/NDCG/MRR prefix curves @1..K from ranked lists (metric.h:17-106).
 * d_truth_ptr int64[B+1], d_truth_items int32 (unique per row). metric_ids host int[n_metrics]
 * (1..5 as in cpp/uni_evaluator.py:14). d_out [B x n_metrics x K]. */
int elimrec_rank_metrics(const int32_t *d_topk_idx, int B, int K, const int64_t *d_truth_ptr,
                         const int32_t *d_truth_items, const int *metric_ids, int n_metrics,
                         float *d_out, void *stream);

/* ---------------------------------------------------------------- pairwise sampler (K20)
 * n triplets: user uniform over the `n_train_users` users with >= 1 training item (with
 * replacement), positive uniform over that user's training items, negative uniform over [0,I)
 * rejecting the user's training items (data/sampler.py:93-126; random_choice.pyx:20-62).
 * Counter-based Philox4x32-10 keyed by (seed, epoch): the stream differs from libc rand() by
 * construction; the contract is distributional. d_user_ids int32[n_train_users]; d_ptr
 * int64[n_train_users+1]; d_items int32 sorted ascending within each user. */
int elimrec_sample_triplets(const int32_t *d_user_ids, const int64_t *d_ptr, const int32_t *d_items,
                            int64_t n_train_users, int64_t I, int64_t n, uint64_t seed,
                            uint64_t epoch, int64_t *d_users, int64_t *d_pos, int64_t *d_neg,
                            void *stream);

/* ================================================================ slab-major propagation (column shards)
 * The d-column table [E_u ; E_i] and every layer table X^k = A X^(k-1) of the folded propagation
 * (models/EliMRec.py:238-248 with the constant feature tables folded out) are kept
 * SLAB-MAJOR: a table of `dl` columns is `ns` slabs of width `w` floats (dl = ns*w, w a power of two >= 4),
 * slab s = a contiguous [n x w] array holding columns [s*w, (s+1)*w) of every row:
 *      element (row r, column s*w + c)  at  d_X[(s*n + r)*w + c].
 * Why: a LightGCN hop is independent per column, so a column slice is a unit of work that needs no
 * communication -- across GPUs (rank q owns dl = d/world columns of the table, of its gradient and of its
 * Adam moments; only the rows of the batch's active nodes ever cross xGMI) and across the 8 XCDs of one GPU
 * (workgroups with equal blockIdx % gs work on the same slab group, whose [n x w*spg] slice fits that
 * XCD's 4 MiB L2 at the Tiktok shape instead of every XCD dragging the whole table through its L2).
 *
 * The adjacency is handed over in SELL-64 work-item form (built once on the host, elimrec_amd/slab.py):
 * work items = the rows with <= long_threshold non-zeros plus <= long_threshold-nnz SEGMENTS of the longer
 * rows, sorted by decreasing length; items are grouped in super blocks of 64 whose (col, val) are stored
 * transposed, entry (block b, neighbour j, item i) at (blk_off[b] + j)*64 + i, so that the lane groups of a
 * wave read their neighbour indices with one coalesced load per step whatever the number of lanes per item.
 * Segment items come first (n_seg_items of them, padded to a multiple of 64); they write partial rows which a
 * second launch adds per long row in a fixed order. Every sum has a fixed order: bitwise reproducible. */
typedef struct elimrec_sell {
    int64_t n_rows;                 /* output rows                                                      */
    int64_t n_src;                  /* rows of the gathered table (= n_rows for the square adjacency)   */
    int32_t n_items;                /* work items incl. padding, a multiple of 64                       */
    int32_t n_seg_items;            /* leading segment items incl. padding, a multiple of 64            */
    int32_t n_seg;                  /* partial slots (= real segment items)                             */
    int32_t n_long;                 /* split rows                                                       */
    const int32_t *d_item_dst;      /* [n_items] output row | partial slot (segment items) | -1 padding */
    const int32_t *d_item_len;      /* [n_items]                                                        */
    const int32_t *d_blk_off;       /* [n_items/64 + 1] first neighbour row of each super block         */
    const int32_t *d_col;           /* SELL column indices                                              */
    const float *d_val;             /* SELL values                                                      */
    const int32_t *d_long_rows;     /* [n_long] ascending                                               */
    const int32_t *d_long_seg_ptr;  /* [n_long + 1] partial slots of each split row                     */
    const int32_t *d_long_index;    /* [n_rows] index into d_long_rows, -1 for unsplit rows             */
    const int64_t *d_rowptr;        /* plain CSR of the same matrix (row-list evaluation); 64-bit: a graph of >= 2^31
                                     * non-zeros (BASELINE.json configs[4]: 2e9) keeps int32 NODE ids but not int32 offsets */
    const int32_t *d_csr_col;
    const float *d_csr_val;
    const int32_t *d_item_long;     /* [n_seg_items] split row (index into d_long_rows) of a segment item; NULL
                                       selects the two-launch form (hop + fix-up) instead of the in-launch combine */
    /* tiered plan (tiered != 0; ONE launch per hop over WAVE TILES, csrc/slab.hip form 3): a tile is the work of one
     * wave of tile_groups lane groups, its index stored [step][group] in d_tile_col / d_tile_val from entry
     * d_tile_off[tile] (int64 [n_tiles + 1], multiples of tile_groups; both arrays padded by 64 entries), with
     * d_tile_len / d_tile_dst [n_tiles x tile_groups] = neighbours of each lane group and its output row (partial
     * slot for segment tiles, -1 padding). Tile order: n_t4 = 4*n_w4 tiles of the n_w4 rows that get a workgroup
     * (contiguous quarters, d_tile_dst = the row), n_t1 tiles of the rows that get a wave (neighbours dealt round-robin
     * to the groups), n_tseg tiles of long_threshold-long segments of the rows longer still (d_tile_long
     * [n_tseg x tile_groups] = index into d_long_rows; partial rows combined in-launch by the last-arriving wave),
     * n_tfin tiles of tile_groups unsplit rows each. n_t1, n_tseg, n_tfin are multiples of 4 (empty padding tiles).
     * d_long_rows / d_long_index / n_long cover every row above long_threshold (compact seg_only output);
     * d_long_seg_ptr / n_seg only the segmented rows' slots. The SELL-64 arrays above may be NULL for a tiered plan. */
    int32_t tiered, n_w1, n_w4;
    int32_t tile_groups, n_t4, n_t1, n_tseg, n_tfin;
    int32_t tile_kmax;              /* most 64-entry index lines of any tile (row stride of the masked hop's bit words) */
    const int64_t *d_tile_off;      /* int64 [n_tiles + 1] */
    const int32_t *d_tile_len, *d_tile_dst, *d_tile_long, *d_tile_col;
    const float *d_tile_val;
} elimrec_sell;

/* The wave-tile plan above built ON THE DEVICE from a device CSR (csrc/plan.hip): the arrays slab.SellPlan builds on the host
 * with numpy sorts over all non-zeros, bit for bit, in three stages with two host read-backs of a few counts in between (the
 * caller sizes the next stage's arrays from them). Tiered plans only. T: rows up to T neighbours are short (G of them per tile);
 * up to T1 a wave per row; up to T2 a workgroup per row (four tiles); longer rows are cut into segments of TS. side_split: rows
 * below are the user side (inside the short rows' tiles the item side goes first; < 0: none); rows_from: rows below take no
 * part (SweepPlan's tile plan of the item rows). Replaces the host-side plan build for graphs a host cannot plan in reasonable
 * time or memory (models/EliMRec.py:309-354 at BASELINE.json configs[3] / configs[4] scale).
 *   elimrec_plan_rows    d_order int32 [n_rows] (rows by class, side, descending length), d_long_rows / d_long_seg_ptr int32
 *                        [n_rows + 1] (n_long / n_long + 1 used), d_long_index int32 [n_rows], d_counts int64 [8] on the device:
 *                        rows per class (workgroup, wave, split, short, none), long rows, segments
 *   elimrec_plan_tiles   d_tile_off int64 [n_tiles + 1], d_tile_len / d_tile_dst int32 [n_tiles x G], d_tile_long int32
 *                        [max(n_tseg x G, 1)], the scatter's d_gb int64 / d_gs int32 [n_tiles x G], d_totals int64 [4] on the
 *                        device: entries, entries before the short rows' tiles, most 64-entry lines of a tile
 *   elimrec_plan_scatter d_tile_col / d_tile_val [entries + 128], zero-filled by the caller
 * Workspace: elimrec_plan_workspace(n_rows, n_seg, n_tiles) bytes (stage 1: n_seg = n_tiles = 1). */
size_t elimrec_plan_workspace(int64_t n_rows, int64_t n_seg, int64_t n_tiles);
int64_t elimrec_plan_tile_count(int64_t n_w4, int64_t n_w1, int64_t n_seg, int64_t n_short, int G);
int elimrec_plan_rows(const int64_t *d_rowptr, int64_t n_rows, int T, int T1, int T2, int TS, int G, int64_t side_split,
                      int64_t rows_from, int32_t *d_order, int32_t *d_long_rows, int32_t *d_long_index, int32_t *d_long_seg_ptr,
                      int64_t *d_counts, void *d_workspace, size_t workspace_bytes, void *stream);
int elimrec_plan_tiles(const int64_t *d_rowptr, int64_t n_rows, int T, int T1, int T2, int TS, int G, int64_t side_split,
                       int64_t rows_from, const int32_t *d_order, const int32_t *d_long_rows, const int32_t *d_long_seg_ptr,
                       int64_t n_w4, int64_t n_w1, int64_t n_split, int64_t n_short, int64_t n_long, int64_t n_seg,
                       int64_t *d_tile_off, int32_t *d_tile_len, int32_t *d_tile_dst, int32_t *d_tile_long, int64_t *d_gb,
                       int32_t *d_gs, int64_t *d_totals, void *d_workspace, size_t workspace_bytes, void *stream);
int elimrec_plan_scatter(int64_t n_tiles, int G, const int64_t *d_tile_off, const int64_t *d_gb, const int32_t *d_tile_len,
                         const int32_t *d_gs, const int32_t *d_col, const float *d_val, int32_t *d_tile_col, float *d_tile_val,
                         void *stream);

/* One hop over a slab-major table:  r = A . Xin ;  Xout[row] = (r + [add_mask bit row] Add[row]) * scale.
 * gs = slab groups (1, 2, 4 or 8 dividing ns): a workgroup works on the ns/gs slabs of group blockIdx % gs.
 * d_src_mask (nullable bitmap over the source rows): rows whose bit is clear are zero and are not read (the
 * first adjoint hop gathers from the row-sparse head gradient). d_add / d_add_mask nullable.
 * d_partials: scratch of elimrec_slab_partials_bytes: [ns x n_seg x w] floats for the split rows, the in-launch
 * combine's arrival counters and (tiered plans) the masked hop's per-index-line source bits; zero-filled once. seg_only != 0: only the split rows are
 * evaluated and written COMPACTLY to d_Xout viewed as [ns x n_long x w] (no add/scale) -- the part of hop L
 * that elimrec_slab_rows cannot do inline; d_add_mask (nullable) then is the bitmap of the rows whose sums are WANTED (the
 * batch's rows: models/EliMRec.py:274-281 reads layer L at the batch rows only) -- the others are not evaluated and their
 * compact rows keep what they held (tiered plans; the two-launch form evaluates every split row).
 * Replaces torch.sparse.mm (models/EliMRec.py:244) and its backward for one column slice. */
size_t elimrec_slab_partials_bytes(const elimrec_sell *A, int ns, int w);
/* d_partials must be zero-filled once after allocation (the one-launch tile form keeps its arrival counters there). */
int elimrec_slab_hop(const elimrec_sell *A, int ns, int w, int gs, const float *d_Xin,
                     const uint32_t *d_src_mask, float *d_Xout, const float *d_add,
                     const uint32_t *d_add_mask, float scale, float *d_partials, size_t partials_bytes,
                     int seg_only, void *stream);
/* The same hop for the rows [0, n_sweep) of ONE side of a bipartite graph whose SOURCE side is a table far beyond the caches
 * (the user rows of BASELINE.json configs[3] / configs[4]; models/EliMRec.py:243-247 for one column slice), csrc/sweep.hip:
 * the rows are cut into contiguous blocks (d_block_ptr int32 [parts * passes * bpx + 1], each <= max_block_rows <=
 * elimrec_slab_sweep_lds_rows(w) rows: a block's output pieces stay in one workgroup's LDS), parts = 8 / min(ns, 8) row parts
 * per slab, bpx workgroups per XCD role, every workgroup takes `passes` blocks one after the other. A block's rows are dealt
 * to the workgroup's waves (w / 4 of them); a wave's non-zeros are laid out window by window (windows of the source range that
 * fit L2), row by row inside a window, columns ascending inside a row, every (wave, window) cut at row boundaries into one
 * chunk per lane group (G = 256 / w per wave) and every chunk into 80-byte step records -- uint32 (source row * w / 4) x 8 |
 * fp32 value x 8 | uint16 (row - the block's first row) x 8; a slot without an entry holds source 0, value 0 and row
 * max_block_rows (a dummy row) -- the same number of records for all lane groups of a (wave, window). d_records holds them as
 * [step][lane group], followed by 8 all-empty steps (the kernel's read-ahead); d_slot_ptr int64 [n_blocks * waves + 1] are
 * the waves' stretches in steps. All waves
 * then gather from two or three windows of the source range at a time, and a source piece leaves L2 once per XCD instead of
 * once per neighbour. A row is summed over its neighbours in column order with fmaf (a fixed order). Rows >= n_sweep are not
 * touched: they are elimrec_slab_hop's on a plan of those rows. w = 32 or 16. */
size_t elimrec_slab_sweep_lds_rows(int w);
int elimrec_slab_sweep_hop(const int64_t *d_slot_ptr, const void *d_records,
                           int64_t n_rows, int64_t n_src, const int32_t *d_block_ptr, int parts, int passes, int bpx,
                           int max_block_rows, int ns, int w, const float *d_Xin, float *d_Xout,
                           const float *d_add, const uint32_t *d_add_mask, float scale, void *stream);
/* ... as the adjoint's LAST hop over the swept rows: their sums are the gradient of the fp32 table d_p_in and are consumed by the
 * Adam step (torch.optim.Adam with coupled L2, main.py:101; the arithmetic of elimrec_slab_hop_adam, element for element) in the
 * launch's epilogue -- no gradient table written and read back; d_grad_out nullable (also store the gradient). The rows of the
 * other side are elimrec_slab_hop_adam's on a plan of those rows (the two launches update disjoint rows of the same buffers). */
int elimrec_slab_sweep_hop_adam(const int64_t *d_slot_ptr, const void *d_records, int64_t n_rows, int64_t n_src,
                                const int32_t *d_block_ptr, int parts, int passes, int bpx, int max_block_rows, int ns, int w,
                                const float *d_Xin, float *d_grad_out, const float *d_add, const uint32_t *d_add_mask, float scale,
                                const float *d_p_in, float *d_p_out, float *d_m, float *d_v, float lr, float beta1, float beta2,
                                float eps, float weight_decay, int64_t step /* 1-based */, void *stream);

/* Tiered plans look the source bitmap up once per index entry before a masked hop (one 64-bit word per index line, kept
 * in d_partials). This entry runs that pass alone -- e.g. on a second stream as soon as the batch's active rows are
 * known -- and elimrec_slab_hop with bit 1 of `seg_only` set (seg_only = 2) then skips it. The bitmap passed to both
 * must hold the same bits. */
int elimrec_slab_source_bits(const elimrec_sell *A, int ns, int w, int gs, const uint32_t *d_src_mask,
                             float *d_partials, size_t partials_bytes, void *stream);

/* elimrec_slab_hop (tiered plan, fp32 tables) with one phase of a weight-gradient batch as extra workgroups behind the
 * hop's tiles: phase 0 = the partial launch (what elimrec_linear_bwd_w_batched_merge(defer_reduce) runs first), phase 1 =
 * elimrec_linear_bwd_w_reduce. SparseAddmmBackward's products and AddmmBackward's dW share launches -- neither reads
 * what the other writes; the weight gradients are off the step's critical path. Same bits as the separate calls.
 * descs / workspace as for elimrec_linear_bwd_w_batched; phase 1 must follow phase 0 on the stream. */
int elimrec_slab_hop_bwd_w(const elimrec_sell *A, int ns, int w, int gs, const float *d_Xin, const uint32_t *d_src_mask,
                           float *d_Xout, const float *d_add, const uint32_t *d_add_mask, float scale, float *d_partials,
                           size_t partials_bytes, int flags, const elimrec_linear_bwd_desc *descs /* host array */, int n,
                           void *d_workspace, size_t workspace_bytes, int phase, void *stream);

/* elimrec_slab_hop (tiered plan, no source bitmap) whose output row pieces are not stored but consumed as the GRADIENT
 * of the same slab-major parameter table by an Adam step with coupled L2 (torch.optim.Adam semantics, arithmetic of
 * elimrec_adam_multi element for element): p_out = adam(p_in, g, m, v) with m, v updated in place; d_p_in / d_p_out
 * may alias or be the two buffers of a ping-pong pair. d_grad_out nullable: also store the gradient there.
 * tail_jobs (nullable, <= 8): further optimizer spans as elimrec_adam_multi takes them -- the projection weights, whose
 * gradients do not depend on this hop -- run by extra workgroups of the same launch.
 * d_sum_src (nullable): one more workgroup leaves sum(d_sum_src[0 .. sum_n)) in d_sum_dst[0], added in elimrec_sum's order
 * (same bits) -- the step's loss from the BPR head's loss rows: it is needed by the host only, so it rides here instead of
 * ending the BPR launch.
 * The adjoint's last hop and the optimizer step (main.py:101) in one launch. */
struct elimrec_adam_job;          /* defined with elimrec_adam_multi below */
int elimrec_slab_hop_adam(const elimrec_sell *A, int ns, int w, int gs, const float *d_Xin, float *d_grad_out,
                          const float *d_add, const uint32_t *d_add_mask, float scale, float *d_partials,
                          size_t partials_bytes, const float *d_p_in, float *d_p_out, float *d_m, float *d_v, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int64_t step,
                          const struct elimrec_adam_job *tail_jobs, int n_tail_jobs, const float *d_sum_src, int64_t sum_n,
                          float *d_sum_dst, void *stream);

/* Layer means (models/EliMRec.py:246-247) of the folded propagation at a list of rows, from slab-major layer
 * tables X^0..X^L (host array of L+1 device pointers):
 *      out0[s]  = 1/(L+1) * (((X^0 + X^1) + X^2) + ... + X^L)[row_s]
 *      narrow[] = 1/(L+1) * sum over even k (user rows, row < U) / odd k (item rows) of X^k[row_s]
 * (the part of Out every feature table shares). layers[L] may be NULL: hop L is then evaluated inline at the
 * listed rows from X^(L-1) through the plain CSR, split rows taken from d_long [ns x n_long x w]
 * (elimrec_slab_hop with seg_only). Rows: d_rows int32 [n_lists x R] with d_counts int32[n_lists] valid
 * entries per list (device; NULL: every entry with a non-negative id is valid -- gathered lists are padded with
 * negative keys), or d_rows NULL = all rows 0..R-1 (n_lists = 1, d_counts ignored).
 * Output row of slot s (= list*R + index): d_out0 + s*ld_out0 and d_narrow + (narrow_by_node ? row : s)*ld_narrow,
 * local column c at offset c (row-major within the row). */
int elimrec_slab_rows(const elimrec_sell *A, int ns, int w, int L, int64_t U,
                      const float *const *layers, const float *d_long, const int32_t *d_rows,
                      const int32_t *d_counts, int64_t R, int n_lists, float *d_out0, int64_t ld_out0,
                      float *d_narrow, int64_t ld_narrow, int narrow_by_node, void *stream);

/* Row-major <-> slab-major: columns [col0, col0 + ns*w) of a row-major [n x ld] table. */
int elimrec_slab_from_rows(const float *d_src, int64_t ld, int64_t col0, int64_t n, int ns, int w,
                           float *d_slab, void *stream);
int elimrec_slab_to_rows(const float *d_slab, int64_t n, int ns, int w, float *d_dst, int64_t ld,
                         int64_t col0, void *stream);

/* Adjoint sources of the folded propagation from [H | G] rows (elimrec_source_rows restricted to a column
 * slice: H = block sum of dOut, G = block 0; 2*dl floats per row, dl = ns*w) contributed by `world` ranks:
 * d_rows [world x R x 2*dl], d_keys int32 [world x R] node ids ascending per rank, negative = padding.
 * Rows of the same node are added in rank order (no float atomics). Writes, slab-major and on the active rows
 * only, SrcA = [H_u ; G_i], SrcB = [G_u ; H_i] and the row bitmap d_mask (all ceil(N/32) words written).
 * M = 0: [H | G] rows as above. M >= 1 (one rank owning every column): d_rows are the dOut rows themselves,
 * [world x R x M*dl], and H (sum of the M column blocks) / G (block 0) are formed on the fly.
 * M = -1 (the wide form, csrc/wide.hip): [H | G] rows as for M = 0, but H goes to SrcA and G to SrcB on EVERY row --
 * SrcA / SrcB are the left / right half of one wide adjoint source.
 * Replaces IndexBackward / index_put(accumulate) across ranks ("sparse-grad reduce-scatter"). */
int elimrec_slab_merge_rows(const float *d_rows, const int32_t *d_keys, int world, int64_t R, int64_t U,
                            int64_t I, int ns, int w, int M, float *d_SrcA, float *d_SrcB, uint32_t *d_mask,
                            void *stream);

/* elimrec_adam_step reading the parameters from d_p_in and writing them to d_p_out (may alias): with two
 * parameter buffers used alternately the tables cached by the last forward keep seeing the parameters they
 * were computed from (models/EliMRec.py:98-99 reads tables made BEFORE the last optimizer step). */
int elimrec_adam_step_out(const float *d_p_in, float *d_p_out, const float *d_g, float *d_m, float *d_v,
                          int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                          int64_t step, void *stream);

/* ---------------------------------------------------------------- fused head forward at the batch's rows
 * Everything after the graph for the <= R active rows of a batch in ONE launch (+ a weight-packing launch): the
 * folded feature blocks Out_m = S_m[node] W_m^T + c[node] b_m^T + narrow (models/EliMRec.py:233-236, DESIGN.md
 * section 2), the fused Linear Y_0 = Out W_side^T + b_side (:262-270; user rows are the first seg_info[1] active rows)
 * and the single-modal heads Y_m = Out_m Ws_m^T + bs_m (:146-151). d_act / d_seg_info: elimrec_segment_plan's active
 * rows and counts; d_out0 / d_narrow: compact [R x 64] rows (layer means of the id table, shared part);
 * d_S[m] [N x D[m]], d_c [N]: the folded constants; weights row-major as torch.nn.Linear holds them.
 * d_OutAct [R x ld_out] receives blocks 1..n_mod, and block 0 too when d_out0 is a separate buffer (16-row form; when
 * d_out0 IS block 0 of d_OutAct it is left as it is),
 * d_YAct [R x ld_y] all 1 + n_mod blocks. d_pack: scratch of elimrec_head_pack_floats floats.
 * phase: 0 = pack the weights, then the head; 1 = pack only (the weights change once per optimizer step: a caller can
 * issue this on a second stream under the forward hops); 2 = head only, d_pack holds the packed weights.
 * The head in two launches (16-row form, packed weights in place): 3 = the feature blocks without the shared part,
 * acc + c * b_m into blocks 1.. of d_OutAct -- needs the active rows and the constants only, not d_out0 / d_narrow, so it
 * can run beside the forward hops; 4 = the rest (the shared part added to what phase 3 left, block 0, the fusion and the
 * single-modal heads). 3 then 4 leave the bits of phase 2.
 * recdim must be 64 and the row tiles must fit LDS, else ELIMREC_E_UNSUPPORTED (callers keep the batched GEMMs). */
size_t elimrec_head_pack_floats(int n_mod, const int *D);
size_t elimrec_head_pack_bwd_offset(int n_mod, const int *D);   /* first float of the head BACKWARD's operands */
int elimrec_head_fwd_fused(const int32_t *d_act, const int32_t *d_seg_info, int64_t R, const float *d_out0,
                           int64_t ld_out0, const float *d_narrow, int64_t ld_nar, const float *d_c, int n_mod,
                           const float *const *d_S, const int64_t *ldS, const int *D, const float *const *d_Wm,
                           const float *const *d_bm, const float *d_Wf_user, const float *d_bf_user,
                           const float *d_Wf_item, const float *d_bf_item, const float *const *d_Ws,
                           const float *const *d_bs, float *d_pack, size_t pack_floats, float *d_OutAct,
                           int64_t ld_out, float *d_YAct, int64_t ld_y, int recdim, int phase, void *stream);

/* elimrec_head_fwd_fused with d_out0 / d_narrow read where the column shards' forward exchange left them (SURVEY section 8(e):
 * each rank evaluates models/EliMRec.py:255-261's layer means on its recdim / world columns and sends every peer the rows that
 * peer's triplets name). d_recv = [world][R][out0: dl | narrow: dl] fp32, piece q = rank q's columns [q*dl, (q+1)*dl) of MY R
 * active rows, world * dl = recdim = 64, dl a multiple of 4. Phases 0, 2, 4. The bits of elimrec_peer_cols_to_rows followed by
 * elimrec_head_fwd_fused on its rows, without that pass over the rows. */
int elimrec_head_fwd_fused_peers(const int32_t *d_act, const int32_t *d_seg_info, int64_t R, const float *d_recv, int world,
                                 int64_t dl, const float *d_c, int n_mod, const float *const *d_S, const int64_t *ldS,
                                 const int *D, const float *const *d_Wm, const float *const *d_bm, const float *d_Wf_user,
                                 const float *d_bf_user, const float *d_Wf_item, const float *d_bf_item,
                                 const float *const *d_Ws, const float *const *d_bs, float *d_pack, size_t pack_floats,
                                 float *d_OutAct, int64_t ld_out, float *d_YAct, int64_t ld_y, int recdim, int phase, void *stream);

/* elimrec_head_fwd_fused with the feature constants read from 16-bit storage where they lie (BASELINE.json configs[1] "bf16";
 * models/EliMRec.py:233-236's v/a/t_dense inputs in their folded form): src->d_table holds one packed row per node,
 * [S_1 | .. | S_n | c_hi c_lo] of fp16 (dtype 1) or bf16 (dtype 2) elements, row_elems elements per row (a multiple of 8) -- the
 * layout elimrec_lookup_pack / _unpack use, this rank holding every row. Rows are widened in registers (exactly; c = hi + lo);
 * arithmetic and results are those of elimrec_lookup_unpack(direct) + elimrec_head_fwd_fused on its fp32 rows, without that pass.
 * d_S_out [R x ld_S_out] / d_c_out [R] (nullable): the widened rows of the launch's active rows, written by the phases that
 * read them (0, 2, 3) for the backward half's weight-gradient launches. */
typedef struct elimrec_head_src16 {
    const void *d_table;
    int64_t row_elems;
    int32_t dtype;
    float *d_S_out;
    int64_t ld_S_out;
    float *d_c_out;
} elimrec_head_src16;
int elimrec_head_fwd_fused_src16(const elimrec_head_src16 *src, const int32_t *d_act, const int32_t *d_seg_info, int64_t R,
                                 const float *d_out0, int64_t ld_out0, const float *d_narrow, int64_t ld_nar, int n_mod,
                                 const int *D, const float *const *d_Wm, const float *const *d_bm, const float *d_Wf_user,
                                 const float *d_bf_user, const float *d_Wf_item, const float *d_bf_item,
                                 const float *const *d_Ws, const float *const *d_bs, float *d_pack, size_t pack_floats,
                                 float *d_OutAct, int64_t ld_out, float *d_YAct, int64_t ld_y, int recdim, int phase, void *stream);

/* Phase 4 of elimrec_head_fwd_fused with the layer means of the active rows EVALUATED by the same launch instead of read back
 * from elimrec_slab_rows (one rank owning every table column, recdim 64 = ns * w): the arguments of that call in a host
 * struct -- plan, slab geometry, layer tables X^0 .. X^L (layers[L] NULL: hop L inline through the plain CSR, split rows
 * from d_long), and where the shared part of the rows is kept (d_narrow_out [R x ld_narrow_out], as elimrec_slab_rows
 * leaves it). Block 0 of d_OutAct is written here. Same bits as elimrec_slab_rows followed by phase 4. */
typedef struct elimrec_head_rows {
    const elimrec_sell *A;
    int32_t ns, w, L;
    int64_t U;
    const float *layers[9];         /* L + 1 device pointers */
    const float *d_long;
    float *d_narrow_out;
    int64_t ld_narrow_out;
} elimrec_head_rows;
int elimrec_head_fwd_fused_rows(const elimrec_head_rows *rows, const int32_t *d_act, const int32_t *d_seg_info, int64_t R,
                                const float *d_c, int n_mod, const float *const *d_S, const int64_t *ldS, const int *D,
                                const float *const *d_Wm, const float *const *d_bm, const float *d_Wf_user,
                                const float *d_bf_user, const float *d_Wf_item, const float *d_bf_item,
                                const float *const *d_Ws, const float *const *d_bs, float *d_pack, size_t pack_floats,
                                float *d_OutAct, int64_t ld_out, float *d_YAct, int64_t ld_y, int recdim, void *stream);

/* ---------------------------------------------------------------- propagation matrix on the device (N3)
 * create_adj_mat (models/EliMRec.py:309-354) from the UNIQUE training interactions d_users / d_items [E] (int64):
 * CSR of plain A (adj_type 0), D^-1/2 A D^-1/2 (1, 'pre'), D^-1 A (2, 'gcmc'), (D+I)^-1 (A+I) (3, 'norm'),
 * D^-1 A + I (4, the reference's fall-through branch); N = U + I rows, columns sorted inside a row, values
 * bit-identical to scipy's: d_pow_table[k] (host-computed with numpy: float32(k^-1/2), float32(k^-1), ...; 0 for
 * k = 0) supplies the only inexact step. d_rowptr [N+1], d_col / d_val [2E (+ N with a diagonal)].
 * *d_err: bit 0 = duplicate interaction, bit 1 = degree beyond the table. */
size_t elimrec_build_adj_workspace(int64_t E, int64_t N, int with_diag);
int elimrec_build_adj(const int64_t *d_users, const int64_t *d_items, int64_t E, int64_t U, int64_t I, int adj_type,
                      const float *d_pow_table, int table_len, int32_t *d_rowptr, int32_t *d_col, float *d_val,
                      int32_t *d_err, void *d_workspace, size_t workspace_bytes, void *stream);

/* The optimizer step of a training step in ONE launch (torch.optim.Adam with coupled L2, main.py:49,101): up to 8 jobs.
 * A job with d_g updates n parameters read from d_p_in and written to d_p_out (may alias); a job without d_g only copies. d_copy_dst (nullable) receives the parameters as they were
 * BEFORE the update (the snapshot of the projection weights the cached tables were computed with,
 * models/EliMRec.py:98-99). `step` is the 1-based step count of the job's parameters. */
typedef struct elimrec_adam_job {
    const float *d_p_in; float *d_p_out;
    const float *d_g; float *d_m; float *d_v;
    float *d_copy_dst;
    int64_t n; int64_t step;
} elimrec_adam_job;
int elimrec_adam_multi(const elimrec_adam_job *jobs /* host array */, int n_jobs, float lr, float beta1, float beta2,
                       float eps, float weight_decay, void *stream);

/* ---------------------------------------------------------------- row-sharded constant tables: id lookup (csrc/lookup.hip)
 * The row partition of SURVEY.md 8(e) for the tables that are only READ at the batch's active rows: the propagated form
 * S_m = mean_k A^k [0 ; F_m] of the V/A/T feature tables (/root/reference/models/EliMRec.py:233-236,366-381) and
 * c = mean_k A^k [0 ; 1]. Rank o owns users [ub[o], ub[o+1]) and items [ib[o], ib[o+1]) (host arrays of world+1 entries,
 * ub[0] = ib[0] = 0, ub[world] = U, ib[world] = I), stored as one local table [own users ; own items] x row_bytes:
 * sum_d elements (dtype 0 = fp32, 1 = fp16, 2 = bf16) of the S_m side by side, then c (fp32: one element; 16-bit: hi + lo in
 * two elements), zero padding to a multiple of 16 bytes.
 * d_acts [world x R] int32: every rank's active-row list (ascending node ids, negative padding behind the valid prefix --
 * what elimrec_batch_plan writes, all-gathered).
 *   lookup_counts  d_counts[r * world + o] = rows of rank r's list that rank o owns (the all_to_all split sizes).
 *   lookup_pack    owner `me`: d_send = my rows of list 0, list 1, ... (per list: users ascending, then items ascending);
 *                  d_send_off [world + 1] (nullable) receives the row offsets of the per-requester chunks.
 *   lookup_unpack  requester `me`: d_rows = the chunks received from owner 0, 1, ... -> d_S [R x sum_d] (leading dimension
 *                  ldS) fp32 rows in the order of d_act (= d_acts + me * R) and d_c [R]. direct = 1 (world 1): d_rows is the
 *                  local table itself, nothing was exchanged. Rows behind the valid prefix of d_act are not written. */
#define ELIMREC_MAX_RANKS 16
int elimrec_lookup_counts(const int32_t *d_acts, int world, int64_t R, int64_t U, int64_t I, const int64_t *ub,
                          const int64_t *ib, int32_t *d_counts, void *stream);
int elimrec_lookup_pack(const int32_t *d_acts, int world, int64_t R, int64_t U, int64_t I, const int64_t *ub,
                        const int64_t *ib, int me, const void *d_shard, int64_t row_bytes, void *d_send,
                        int32_t *d_send_off, void *stream);
int elimrec_lookup_unpack(const int32_t *d_act, int world, int64_t R, int64_t U, int64_t I, const int64_t *ub,
                          const int64_t *ib, int me, const void *d_rows, int64_t row_bytes, int dtype, int sum_d,
                          int direct, float *d_S, int64_t ldS, float *d_c, void *stream);

/* ---------------------------------------------------------------- folded propagation for adjacencies with a diagonal (csrc/wide.hip)
 * adj_type = norm / mean + I (/root/reference/models/EliMRec.py:332-335,349-352): the E_u-borne and the E_i-borne part of the
 * tables no longer alternate by layer, so the graph carries both side by side in a WIDE slab table [N x 2 dl] (2 ns slabs of w
 * floats; slabs [0, ns) = left = the E_u-borne part). Layer 0 from the [N x dl] parameters; (layer mean, shared part) of listed
 * rows (d_rows NULL: rows 0 .. total-1; negative ids are padding): out0 = mean_k (left + right), narrow = mean_k left; the
 * parameters' gradient from the wide adjoint table: left half on user rows, right half on item rows, times scale. */
int elimrec_wide_from_master(const float *d_master, int64_t U, int64_t N, int ns, int w, float *d_wide, void *stream);
int elimrec_wide_rows(const float *const *layers /* L+1 wide tables */, int L, int64_t N, int ns, int w, const int32_t *d_rows,
                      int64_t total, float *d_out0, int64_t ld_out0, float *d_narrow, int64_t ld_narrow, void *stream);
int elimrec_wide_grad(const float *d_wide_grad, int64_t U, int64_t N, int ns, int w, float scale, float *d_grad, void *stream);
/* Row bitmap of `world` active-row lists d_keys [world x R] (ascending, negative padding last): bit n of d_mask
 * ((N + 31) / 32 words) <=> node n is in some list -- the words elimrec_slab_merge_rows writes for the same lists. */
int elimrec_rows_bitmap(const int32_t *d_keys, int world, int64_t R, int64_t N, uint32_t *d_mask, void *stream);
/* Column shards, forward exchange: d_recv [world x R x 2 x dl] = per peer q the (layer mean | shared part) of MY R active rows
 * in q's dl columns -> rows: d_out0[r, q*dl + c] (leading dimension ld0) and d_out1[r, q*dl + c] (ld1). */
int elimrec_peer_cols_to_rows(const float *d_recv, int world, int64_t R, int dl, float *d_out0, int64_t ld0, float *d_out1,
                              int64_t ld1, void *stream);

/* ---------------------------------------------------------------- a training step as one host call (csrc/program.hip)
 * The reference's loop body (main.py:98-101) is, on this path, ~13 launches on two HIP streams plus -- with several ranks --
 * four collectives. A program is that sequence written down once: calls of THIS library's entry points with their arguments
 * packed as 64-bit words (pointers, integers, fp32 bit patterns in the low half), event record / wait pairs that hand work
 * between streams, and RCCL collectives on a communicator owned by this library, enqueued on the caller's stream.
 * elimrec_program_run applies `patches` (argument words that change every step: the batch's index tensors, the loss slot,
 * Adam's step count) and issues the list. Nothing here allocates device memory or synchronises the host. */
#define ELIMREC_PROGRAM_MAX_ARGS 32
enum { ELIMREC_OP_CALL = 0,          /* fn = index into the function table (elimrec_program_fn_name), args = its arguments      */
       ELIMREC_OP_RECORD = 1,        /* args: stream, event slot                                                                */
       ELIMREC_OP_WAIT = 2 };        /* args: stream, event slot                                                                */
typedef struct elimrec_op { int32_t kind; int32_t fn; uint64_t args[ELIMREC_PROGRAM_MAX_ARGS]; } elimrec_op;
typedef struct elimrec_patch { int32_t op; int32_t arg; uint64_t value; } elimrec_patch;
int elimrec_program_fn_count(void);
const char *elimrec_program_fn_name(int i);
int elimrec_program_fn_args(int i);
int elimrec_program_create(const elimrec_op *ops, int n_ops, void **prog_out);
/* ... with the scope of the RECORD / WAIT events stated by the caller: system_scope_events != 0 keeps the default (system-scope
 * fence at every record: needed when peers' or the host's writes must be visible behind an event); 0 = fence-free events between
 * the streams of one device, unless the list holds one of this library's RCCL calls (then system scope regardless). */
int elimrec_program_create_scoped(const elimrec_op *ops, int n_ops, int system_scope_events, void **prog_out);
int elimrec_program_run(void *prog, const elimrec_patch *patches, int n_patches);
int elimrec_program_destroy(void *prog);
/* RCCL communicator of the ranks of a job (one per process / GPU): rank 0 draws the 128-byte id, the host side hands it to the
 * others (torch.distributed broadcast), every rank calls elimrec_comm_create. The RCCL library is the one the process has
 * already loaded (PyTorch-ROCm's), resolved at run time. */
int elimrec_comm_unique_id(void *id128);
int elimrec_comm_create(const void *id128, int world, int rank, void **comm_out);
int elimrec_comm_nranks(void *comm, int *n_out);      /* ncclCommCount of the communicator: what a benchmark line reports */
int elimrec_comm_destroy(void *comm);
/* The step's exchanges on that communicator, enqueued on `stream` (sizes in BYTES; all_reduce: fp32 sum in place;
 * all_to_all_v: `sizes` = HOST int64 [2 x world], bytes to send to peer 0.. then bytes to receive from peer 0.., chunks back
 * to back in d_send / d_recv). Callable directly or from a program (they are in its function table). */
int elimrec_comm_all_gather(void *comm, const void *d_send, void *d_recv, int64_t bytes_per_rank, void *stream);
int elimrec_comm_all_reduce_f32(void *comm, float *d_buf, int64_t n, void *stream);
int elimrec_comm_all_to_all(void *comm, const void *d_send, void *d_recv, int64_t bytes_per_peer, void *stream);
int elimrec_comm_all_to_all_v(void *comm, const void *d_send, void *d_recv, const int64_t *sizes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ELIMREC_HIP_H */
