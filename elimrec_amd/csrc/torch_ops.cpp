// torch.ops.elimrec.* : a thin TORCH_LIBRARY registration over the C ABI of libelimrec_hip.so (include/elimrec_hip.h),
// the form SURVEY.md 8(b) names for a PyTorch host. Nothing is computed here: every op checks its tensors, unpacks
// pointers / strides, takes torch's current HIP stream, allocates its outputs and scratch with torch and calls ONE
// entry point of the C ABI; failures become c10::Error (RuntimeError in Python) carrying elimrec_last_error().
// The ctypes binding (elimrec_amd/_lib.py) stays for hosts without torch; both call the same library.
//
//   propagate(rowptr i32[N+1], col i32[nnz], val f32[nnz], X f32[N x C], L, transpose=False) -> mean over the L+1 layer tables
//       (transpose: with A^T -- the adjoint of a propagation matrix that is not self-adjoint)
//   segment_reduce(rows f32[n x ld], keys i32[n], split_key) -> (active keys, summed rows, seg_info): the deterministic
//       index_put(accumulate) of the backward pass; bpr_head_fwd's grad_rows + keys are its inputs (= "bpr_head_bwd")
//   linear_fwd(A, W, bias?) -> A W^T + bias                linear_bwd_w(A, B) -> (A^T B, column sums of A)
//   bpr_head_fwd(Y, U, I, users, pos, neg, d, block_weights) -> (loss_rows, grad_rows, keys)
//   adam_step_(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step) -> p
//   score_topk(Y, U, I, users, d, S, head_mask, fusion_mode, predict_type, train_ptr?, train_items?, K, tie_order=0) -> (idx, val)
//       (tie_order 1: the reference's partial_sort_copy order among equal scores, evaluate.h:26-33, replayed on the device)
//   rank_metrics(topk_idx, truth_ptr, truth_items, metric_ids) -> f32[B x n_metrics x K]
//   sample_triplets(user_ids, ptr, items, num_items, n, seed, epoch) -> (users, pos, neg)
#include <ATen/ATen.h>
#include <ATen/hip/HIPContext.h>
#include <torch/library.h>

#include <vector>

#include "../../include/elimrec_hip.h"

namespace {

void *cur_stream() { return (void *)at::hip::getCurrentHIPStream().stream(); }

void check(int rc, const char *what) {
    if (rc != 0) {
        const char *msg = elimrec_last_error();
        TORCH_CHECK(false, "elimrec::", what, " failed (rc=", rc, "): ", msg ? msg : "?");
    }
}

void need(const at::Tensor &t, const char *name, at::ScalarType dt, int64_t dim = -1) {
    TORCH_CHECK(t.is_cuda(), "elimrec: '", name, "' must be a HIP device tensor (the hot path has no CPU implementation)");
    TORCH_CHECK(t.scalar_type() == dt, "elimrec: '", name, "' has dtype ", t.scalar_type(), ", expected ", dt);
    TORCH_CHECK(dim < 0 || t.dim() == dim, "elimrec: '", name, "' must be ", dim, "-D");
}

const at::Tensor rowmajor(const at::Tensor &t, const char *name) {
    need(t, name, at::kFloat, 2);
    return t.stride(1) == 1 ? t : t.contiguous();
}

at::Tensor propagate(const at::Tensor &rowptr, const at::Tensor &col, const at::Tensor &val, const at::Tensor &X, int64_t L,
                     bool transpose) {
    need(rowptr, "rowptr", at::kInt, 1); need(col, "col", at::kInt, 1); need(val, "val", at::kFloat, 1);
    need(X, "X", at::kFloat, 2);
    const at::Tensor x = X.contiguous();
    at::Tensor rp = rowptr.contiguous(), c = col.contiguous(), v = val.contiguous();
    const int64_t n = x.size(0), C = x.size(1);
    TORCH_CHECK(rp.numel() == n + 1, "elimrec::propagate: rowptr has ", rp.numel(), " entries for ", n, " rows");
    if (transpose) {
        // the adjoint's matrix (SparseAddmmBackward multiplies by A^T; 'pre' is self-adjoint, 'gcmc' / 'norm' are not): CSR of
        // A^T from CSR of A -- index bookkeeping only (a stable sort by column), the propagation itself is the same kernel
        const at::Tensor counts = (rp.slice(0, 1) - rp.slice(0, 0, n)).to(at::kLong);
        const at::Tensor rows = at::repeat_interleave(at::arange(n, counts.options()), counts);
        const at::Tensor cl = c.to(at::kLong);
        const at::Tensor order = at::argsort(cl * n + rows, /*stable=*/true, 0, false);
        rp = at::cat({at::zeros({1}, counts.options()), at::cumsum(at::bincount(cl, {}, n), 0)}).to(at::kInt);
        c = rows.index_select(0, order).to(at::kInt);
        v = v.index_select(0, order);
    }
    at::Tensor out = at::empty_like(x), t0 = at::empty_like(x), t1 = at::empty_like(x);
    check(elimrec_propagate(rp.data_ptr<int32_t>(), c.data_ptr<int32_t>(), v.data_ptr<float>(), n, (int)C, nullptr, (int)L,
                            x.data_ptr<float>(), t0.data_ptr<float>(), t1.data_ptr<float>(), out.data_ptr<float>(), cur_stream()),
          "propagate");
    return out;
}

at::Tensor linear_fwd(const at::Tensor &A, const at::Tensor &W, const c10::optional<at::Tensor> &bias) {
    const at::Tensor a = rowmajor(A, "A"), w = rowmajor(W, "W");
    TORCH_CHECK(a.size(1) == w.size(1), "elimrec::linear_fwd: A is [", a.size(0), " x ", a.size(1), "], W is [", w.size(0), " x ", w.size(1), "]");
    at::Tensor b;
    if (bias.has_value() && bias->defined()) { need(*bias, "bias", at::kFloat, 1); b = bias->contiguous(); }
    at::Tensor out = at::empty({a.size(0), w.size(0)}, a.options());
    check(elimrec_linear_fwd(a.data_ptr<float>(), a.stride(0), w.data_ptr<float>(), w.stride(0), b.defined() ? b.data_ptr<float>() : nullptr,
                             out.data_ptr<float>(), out.stride(0), a.size(0), (int)w.size(0), (int)a.size(1), cur_stream()),
          "linear_fwd");
    return out;
}

std::tuple<at::Tensor, at::Tensor> linear_bwd_w(const at::Tensor &A, const at::Tensor &B) {
    const at::Tensor a = rowmajor(A, "A"), b = rowmajor(B, "B");
    TORCH_CHECK(a.size(0) == b.size(0), "elimrec::linear_bwd_w: row counts differ");
    at::Tensor out = at::empty({a.size(1), b.size(1)}, a.options()), colsum = at::empty({a.size(1)}, a.options());
    const size_t need_ws = elimrec_linear_bwd_w_workspace(a.size(0), (int)a.size(1), (int)b.size(1));
    at::Tensor ws = at::empty({(int64_t)(need_ws ? need_ws : 1)}, a.options().dtype(at::kByte));
    check(elimrec_linear_bwd_w(a.data_ptr<float>(), a.stride(0), b.data_ptr<float>(), b.stride(0), nullptr, nullptr, a.size(0),
                               (int)a.size(1), (int)b.size(1), out.data_ptr<float>(), out.stride(0), colsum.data_ptr<float>(), 0,
                               ws.data_ptr(), (size_t)ws.numel(), cur_stream()),
          "linear_bwd_w");
    return {out, colsum};
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> bpr_head_fwd(const at::Tensor &Y, int64_t U, int64_t I, const at::Tensor &users,
                                                            const at::Tensor &pos, const at::Tensor &neg, int64_t d,
                                                            std::vector<double> block_weights) {
    const at::Tensor y = rowmajor(Y, "Y");
    need(users, "users", at::kLong, 1); need(pos, "pos", at::kLong, 1); need(neg, "neg", at::kLong, 1);
    const int64_t B = users.numel();
    TORCH_CHECK(pos.numel() == B && neg.numel() == B, "elimrec::bpr_head_fwd: users / pos / neg differ in length");
    TORCH_CHECK(y.size(0) == U + I && y.size(1) == d * (int64_t)block_weights.size(), "elimrec::bpr_head_fwd: Y must be [(U+I) x d*blocks]");
    std::vector<float> w(block_weights.begin(), block_weights.end());
    at::Tensor loss = at::empty({B}, y.options()), grad = at::empty({3 * B, y.size(1)}, y.options());
    at::Tensor keys = at::empty({3 * B}, y.options().dtype(at::kInt));
    const at::Tensor u = users.contiguous(), p = pos.contiguous(), n = neg.contiguous();
    check(elimrec_bpr_head(y.data_ptr<float>(), y.stride(0), U, I, u.data_ptr<int64_t>(), p.data_ptr<int64_t>(), n.data_ptr<int64_t>(),
                           (int)B, (int)d, (int)w.size(), w.data(), loss.data_ptr<float>(), grad.data_ptr<float>(),
                           keys.data_ptr<int32_t>(), cur_stream()),
          "bpr_head_fwd");
    return {loss, grad, keys};
}

at::Tensor adam_step_(at::Tensor p, const at::Tensor &g, at::Tensor m, at::Tensor v, double lr, double beta1, double beta2,
                      double eps, double weight_decay, int64_t step) {
    need(p, "p", at::kFloat); need(g, "g", at::kFloat); need(m, "m", at::kFloat); need(v, "v", at::kFloat);
    TORCH_CHECK(p.is_contiguous() && g.is_contiguous() && m.is_contiguous() && v.is_contiguous(), "elimrec::adam_step_: contiguous tensors");
    TORCH_CHECK(g.numel() == p.numel() && m.numel() == p.numel() && v.numel() == p.numel(), "elimrec::adam_step_: sizes differ");
    check(elimrec_adam_step(p.data_ptr<float>(), g.data_ptr<float>(), m.data_ptr<float>(), v.data_ptr<float>(), p.numel(), (float)lr,
                            (float)beta1, (float)beta2, (float)eps, (float)weight_decay, step, cur_stream()),
          "adam_step_");
    return p;
}

std::tuple<at::Tensor, at::Tensor> score_topk(const at::Tensor &Y, int64_t U, int64_t I, const at::Tensor &users, int64_t d, int64_t S,
                                              int64_t head_mask, int64_t fusion_mode, int64_t predict_type,
                                              const c10::optional<at::Tensor> &train_ptr,
                                              const c10::optional<at::Tensor> &train_items, int64_t K, int64_t tie_order) {
    const at::Tensor y = rowmajor(Y, "Y");
    need(users, "users", at::kLong, 1);
    const at::Tensor u = users.contiguous();
    const int64_t B = u.numel();
    TORCH_CHECK(K >= 1, "elimrec::score_topk: K >= 1");
    TORCH_CHECK(tie_order == 0 || tie_order == 1, "elimrec::score_topk: tie_order 0 (score desc, id asc) or 1 (the reference's heap order)");
    at::Tensor tp, ti;
    if (train_ptr.has_value() && train_ptr->defined()) {
        TORCH_CHECK(train_items.has_value() && train_items->defined(), "elimrec::score_topk: train_ptr needs train_items");
        need(*train_ptr, "train_ptr", at::kLong, 1); need(*train_items, "train_items", at::kInt, 1);
        tp = train_ptr->contiguous(); ti = train_items->contiguous();
    }
    at::Tensor idx = at::empty({B, K}, y.options().dtype(at::kInt)), val = at::empty({B, K}, y.options());
    const size_t need_ws = elimrec_score_workspace2((int)B, U, I, (int)S, (int)K);
    at::Tensor ws = at::empty({(int64_t)(need_ws ? need_ws : 1)}, y.options().dtype(at::kByte));
    check(elimrec_score_topk_ordered(y.data_ptr<float>(), y.stride(0), U, I, u.data_ptr<int64_t>(), (int)B, (int)d, (int)S, (uint32_t)head_mask,
                                     (int)fusion_mode, (int)predict_type, nullptr, tp.defined() ? tp.data_ptr<int64_t>() : nullptr,
                                     ti.defined() ? ti.data_ptr<int32_t>() : nullptr, nullptr, 0, (int)K, idx.data_ptr<int32_t>(),
                                     val.data_ptr<float>(), ws.data_ptr(), (size_t)ws.numel(), (int)tie_order, cur_stream()),
          "score_topk");
    return {idx, val};
}

at::Tensor rank_metrics(const at::Tensor &topk_idx, const at::Tensor &truth_ptr, const at::Tensor &truth_items,
                        std::vector<int64_t> metric_ids) {
    need(topk_idx, "topk_idx", at::kInt, 2); need(truth_ptr, "truth_ptr", at::kLong, 1); need(truth_items, "truth_items", at::kInt, 1);
    const at::Tensor t = topk_idx.contiguous(), tp = truth_ptr.contiguous(), ti = truth_items.contiguous();
    std::vector<int> ids(metric_ids.begin(), metric_ids.end());
    at::Tensor out = at::empty({t.size(0), (int64_t)ids.size() * t.size(1)}, t.options().dtype(at::kFloat));
    check(elimrec_rank_metrics(t.data_ptr<int32_t>(), (int)t.size(0), (int)t.size(1), tp.data_ptr<int64_t>(), ti.data_ptr<int32_t>(),
                               ids.data(), (int)ids.size(), out.data_ptr<float>(), cur_stream()),
          "rank_metrics");
    return out;
}

// IndexBackward / index_put(accumulate): rows with equal keys summed in ascending row order, deterministic.
// -> (active keys int32[n] (first n_active valid), reduced f32[n x ld], seg_info int32[8] = {n_active, #keys < split_key, ...})
std::tuple<at::Tensor, at::Tensor, at::Tensor> segment_reduce(const at::Tensor &rows, const at::Tensor &keys, int64_t split_key) {
    need(keys, "keys", at::kInt, 1);
    const at::Tensor r = rowmajor(rows, "rows").contiguous(), k = keys.contiguous();
    const int64_t n = r.size(0), ld = r.size(1);
    TORCH_CHECK(k.numel() == n, "elimrec::segment_reduce: one key per row");
    at::Tensor active = at::empty({n}, k.options()), reduced = at::empty({n, ld}, r.options());
    at::Tensor seg = at::zeros({8}, k.options());
    const size_t need_ws = elimrec_segment_reduce_workspace(n);
    at::Tensor ws = at::empty({(int64_t)(need_ws ? need_ws : 1)}, r.options().dtype(at::kByte));
    check(elimrec_segment_reduce_rows(r.data_ptr<float>(), k.data_ptr<int32_t>(), n, (int)ld, (int32_t)split_key, active.data_ptr<int32_t>(),
                                      reduced.data_ptr<float>(), nullptr, seg.data_ptr<int32_t>(), ws.data_ptr(), (size_t)ws.numel(),
                                      cur_stream()),
          "segment_reduce");
    return {active, reduced, seg};
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> sample_triplets(const at::Tensor &user_ids, const at::Tensor &ptr, const at::Tensor &items,
                                                               int64_t num_items, int64_t n, int64_t seed, int64_t epoch) {
    need(user_ids, "user_ids", at::kInt, 1); need(ptr, "ptr", at::kLong, 1); need(items, "items", at::kInt, 1);
    const at::Tensor uid = user_ids.contiguous(), p = ptr.contiguous(), it = items.contiguous();
    at::Tensor u = at::empty({n}, p.options()), po = at::empty({n}, p.options()), ne = at::empty({n}, p.options());
    check(elimrec_sample_triplets(uid.data_ptr<int32_t>(), p.data_ptr<int64_t>(), it.data_ptr<int32_t>(), uid.numel(), num_items, n,
                                  (uint64_t)seed, (uint64_t)epoch, u.data_ptr<int64_t>(), po.data_ptr<int64_t>(), ne.data_ptr<int64_t>(),
                                  cur_stream()),
          "sample_triplets");
    return {u, po, ne};
}

// "bpr_head_bwd" of SURVEY.md 8(b): the row gradients bpr_head_fwd saved, scaled by the upstream gradient of the mean loss and
// reduced per node in ascending slot order (IndexBackward, deterministic) into a dense dY [n_rows x Cy].
at::Tensor bpr_head_bwd(const at::Tensor &grad_rows, const at::Tensor &keys, const at::Tensor &grad_out, int64_t n_rows) {
    need(keys, "keys", at::kInt, 1); need(grad_out, "grad_out", at::kFloat);
    TORCH_CHECK(grad_out.numel() == 1, "elimrec::bpr_head_bwd: grad_out is the gradient of the scalar loss");
    const at::Tensor r = rowmajor(grad_rows, "grad_rows").contiguous(), k = keys.contiguous();
    const int64_t n = r.size(0), ld = r.size(1);
    TORCH_CHECK(k.numel() == n && n_rows >= 1, "elimrec::bpr_head_bwd: one key per row");
    at::Tensor active = at::empty({n}, k.options()), reduced = at::empty({n, ld}, r.options()), seg = at::zeros({8}, k.options());
    const size_t need_ws = elimrec_segment_reduce_workspace(n);
    at::Tensor ws = at::empty({(int64_t)(need_ws ? need_ws : 1)}, r.options().dtype(at::kByte));
    check(elimrec_segment_reduce_rows(r.data_ptr<float>(), k.data_ptr<int32_t>(), n, (int)ld, 0, active.data_ptr<int32_t>(),
                                      reduced.data_ptr<float>(), nullptr, seg.data_ptr<int32_t>(), ws.data_ptr(), (size_t)ws.numel(),
                                      cur_stream()),
          "bpr_head_bwd");
    // rows behind the valid prefix go to a dump row (no host read-back of the count)
    const at::Tensor slot = at::arange(n, k.options());
    const at::Tensor dst = at::where(slot < seg[0], active, at::full({}, n_rows, k.options())).to(at::kLong);
    at::Tensor dY = at::zeros({n_rows + 1, ld}, r.options());
    dY.index_copy_(0, dst, reduced * grad_out.reshape({}));
    return dY.narrow(0, 0, n_rows);
}

// ---- item-sharded evaluation (elimrec_score_topk_shard phases 1 / 2, elimrec_topk_merge)
static void train_csr(const c10::optional<at::Tensor> &train_ptr, const c10::optional<at::Tensor> &train_items, at::Tensor &tp, at::Tensor &ti) {
    if (train_ptr.has_value() && train_ptr->defined()) {
        TORCH_CHECK(train_items.has_value() && train_items->defined(), "elimrec: train_ptr needs train_items");
        need(*train_ptr, "train_ptr", at::kLong, 1); need(*train_items, "train_items", at::kInt, 1);
        tp = train_ptr->contiguous(); ti = train_items->contiguous();
    }
}

at::Tensor score_shard_row_sums(const at::Tensor &Y, int64_t U, int64_t I, const at::Tensor &users, int64_t d, int64_t S, int64_t head_mask,
                                int64_t fusion_mode, int64_t I_total) {
    const at::Tensor y = rowmajor(Y, "Y");
    need(users, "users", at::kLong, 1);
    const at::Tensor u = users.contiguous();
    const int64_t B = u.numel();
    at::Tensor row_sum = at::zeros({B}, y.options());
    const size_t need_ws = elimrec_score_workspace_for((int)B, U, I, (int)S, 1, (int)d, 0);
    at::Tensor ws = at::empty({(int64_t)(need_ws ? need_ws : 1)}, y.options().dtype(at::kByte));
    check(elimrec_score_topk_shard(y.data_ptr<float>(), y.stride(0), U, I, u.data_ptr<int64_t>(), (int)B, (int)d, (int)S, (uint32_t)head_mask,
                                   (int)fusion_mode, 2, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, ws.data_ptr(),
                                   (size_t)ws.numel(), 1, row_sum.data_ptr<float>(), I_total, 0, cur_stream()),
          "score_shard_row_sums");
    return row_sum;
}

std::tuple<at::Tensor, at::Tensor> score_topk_shard(const at::Tensor &Y, int64_t U, int64_t I, const at::Tensor &users, int64_t d, int64_t S,
                                                    int64_t head_mask, int64_t fusion_mode, int64_t predict_type,
                                                    const c10::optional<at::Tensor> &train_ptr, const c10::optional<at::Tensor> &train_items,
                                                    int64_t K, const at::Tensor &row_sum, int64_t I_total, int64_t id_offset) {
    const at::Tensor y = rowmajor(Y, "Y");
    need(users, "users", at::kLong, 1); need(row_sum, "row_sum", at::kFloat, 1);
    const at::Tensor u = users.contiguous(), rs = row_sum.contiguous();
    const int64_t B = u.numel();
    TORCH_CHECK(K >= 1 && rs.numel() == B, "elimrec::score_topk_shard: K >= 1, one row sum per user");
    at::Tensor tp, ti;
    train_csr(train_ptr, train_items, tp, ti);
    at::Tensor idx = at::empty({B, K}, y.options().dtype(at::kInt)), val = at::empty({B, K}, y.options());
    const size_t need_ws = elimrec_score_workspace_for((int)B, U, I, (int)S, (int)K, (int)d, 0);
    at::Tensor ws = at::empty({(int64_t)(need_ws ? need_ws : 1)}, y.options().dtype(at::kByte));
    check(elimrec_score_topk_shard(y.data_ptr<float>(), y.stride(0), U, I, u.data_ptr<int64_t>(), (int)B, (int)d, (int)S, (uint32_t)head_mask,
                                   (int)fusion_mode, (int)predict_type, nullptr, tp.defined() ? tp.data_ptr<int64_t>() : nullptr,
                                   ti.defined() ? ti.data_ptr<int32_t>() : nullptr, nullptr, 0, (int)K, idx.data_ptr<int32_t>(),
                                   val.data_ptr<float>(), ws.data_ptr(), (size_t)ws.numel(), 2, rs.data_ptr<float>(), I_total, id_offset,
                                   cur_stream()),
          "score_topk_shard");
    return {idx, val};
}

std::tuple<at::Tensor, at::Tensor> topk_merge(const at::Tensor &cand_val, const at::Tensor &cand_idx, int64_t K) {
    need(cand_val, "cand_val", at::kFloat, 2); need(cand_idx, "cand_idx", at::kInt, 2);
    const at::Tensor v = cand_val.contiguous(), i = cand_idx.contiguous();
    TORCH_CHECK(v.sizes() == i.sizes() && K >= 1 && K <= v.size(1), "elimrec::topk_merge: [B x n] values and ids, 1 <= K <= n");
    at::Tensor idx = at::empty({v.size(0), K}, i.options()), val = at::empty({v.size(0), K}, v.options());
    check(elimrec_topk_merge(v.data_ptr<float>(), i.data_ptr<int32_t>(), (int)v.size(0), (int)v.size(1), (int)K, idx.data_ptr<int32_t>(),
                             val.data_ptr<float>(), cur_stream()),
          "topk_merge");
    return {idx, val};
}

// ---- row-sharded feature constants: the all-to-all id lookup (elimrec_lookup_counts / _pack / _unpack). user_bounds / item_bounds:
// world + 1 ascending row bounds of the owners' blocks.
static void bounds(const std::vector<int64_t> &ub, const std::vector<int64_t> &ib, int64_t &world) {
    world = (int64_t)ub.size() - 1;
    TORCH_CHECK(world >= 1 && world <= ELIMREC_MAX_RANKS && ib.size() == ub.size(), "elimrec::lookup: user_bounds / item_bounds hold world + 1 entries");
}

at::Tensor lookup_counts(const at::Tensor &acts, int64_t U, int64_t I, std::vector<int64_t> user_bounds, std::vector<int64_t> item_bounds) {
    int64_t world;
    bounds(user_bounds, item_bounds, world);
    need(acts, "acts", at::kInt, 2);
    const at::Tensor a = acts.contiguous();
    TORCH_CHECK(a.size(0) == world, "elimrec::lookup_counts: one active-row list per rank");
    at::Tensor counts = at::empty({world, world}, a.options());
    check(elimrec_lookup_counts(a.data_ptr<int32_t>(), (int)world, a.size(1), U, I, user_bounds.data(), item_bounds.data(),
                                counts.data_ptr<int32_t>(), cur_stream()),
          "lookup_counts");
    return counts;
}

std::tuple<at::Tensor, at::Tensor> lookup_pack(const at::Tensor &acts, int64_t U, int64_t I, std::vector<int64_t> user_bounds,
                                               std::vector<int64_t> item_bounds, int64_t me, const at::Tensor &shard, int64_t row_bytes) {
    int64_t world;
    bounds(user_bounds, item_bounds, world);
    need(acts, "acts", at::kInt, 2);
    const at::Tensor a = acts.contiguous();
    TORCH_CHECK(shard.is_cuda() && shard.is_contiguous() && row_bytes % 16 == 0 && me >= 0 && me < world && a.size(0) == world,
                "elimrec::lookup_pack: contiguous device shard, rows padded to 16 bytes, 0 <= me < world");
    at::Tensor send = at::empty({world * a.size(1), row_bytes}, a.options().dtype(at::kByte));     // worst case: every listed row is mine
    at::Tensor off = at::empty({world + 1}, a.options());
    check(elimrec_lookup_pack(a.data_ptr<int32_t>(), (int)world, a.size(1), U, I, user_bounds.data(), item_bounds.data(), (int)me,
                              shard.data_ptr(), row_bytes, send.data_ptr(), off.data_ptr<int32_t>(), cur_stream()),
          "lookup_pack");
    return {send, off};
}

std::tuple<at::Tensor, at::Tensor> lookup_unpack(const at::Tensor &act, int64_t U, int64_t I, std::vector<int64_t> user_bounds,
                                                 std::vector<int64_t> item_bounds, int64_t me, const at::Tensor &rows, int64_t row_bytes,
                                                 int64_t dtype, int64_t sum_d, bool direct) {
    int64_t world;
    bounds(user_bounds, item_bounds, world);
    need(act, "act", at::kInt, 1);
    const at::Tensor a = act.contiguous();
    TORCH_CHECK(rows.is_cuda() && rows.is_contiguous() && sum_d >= 1 && dtype >= 0 && dtype <= 2, "elimrec::lookup_unpack: bad arguments");
    at::Tensor S = at::zeros({a.numel(), sum_d}, a.options().dtype(at::kFloat)), c = at::zeros({a.numel()}, a.options().dtype(at::kFloat));
    check(elimrec_lookup_unpack(a.data_ptr<int32_t>(), (int)world, a.numel(), U, I, user_bounds.data(), item_bounds.data(), (int)me,
                                rows.data_ptr(), row_bytes, (int)dtype, (int)sum_d, direct ? 1 : 0, S.data_ptr<float>(), S.stride(0),
                                c.data_ptr<float>(), cur_stream()),
          "lookup_unpack");
    return {S, c};
}

}  // namespace

TORCH_LIBRARY(elimrec, m) {
    m.def("propagate(Tensor rowptr, Tensor col, Tensor val, Tensor X, int L, bool transpose=False) -> Tensor");
    m.def("segment_reduce(Tensor rows, Tensor keys, int split_key) -> (Tensor, Tensor, Tensor)");
    m.def("linear_fwd(Tensor A, Tensor W, Tensor? bias) -> Tensor");
    m.def("linear_bwd_w(Tensor A, Tensor B) -> (Tensor, Tensor)");
    m.def("bpr_head_fwd(Tensor Y, int U, int I, Tensor users, Tensor pos, Tensor neg, int d, float[] block_weights) -> (Tensor, Tensor, Tensor)");
    m.def("adam_step_(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, float lr, float beta1, float beta2, float eps, float weight_decay, int step) -> Tensor(a!)");
    m.def("score_topk(Tensor Y, int U, int I, Tensor users, int d, int S, int head_mask, int fusion_mode, int predict_type, Tensor? train_ptr, Tensor? train_items, int K, int tie_order=0) -> (Tensor, Tensor)");
    m.def("rank_metrics(Tensor topk_idx, Tensor truth_ptr, Tensor truth_items, int[] metric_ids) -> Tensor");
    m.def("sample_triplets(Tensor user_ids, Tensor ptr, Tensor items, int num_items, int n, int seed, int epoch) -> (Tensor, Tensor, Tensor)");
    m.def("bpr_head_bwd(Tensor grad_rows, Tensor keys, Tensor grad_out, int n_rows) -> Tensor");
    m.def("score_shard_row_sums(Tensor Y, int U, int I, Tensor users, int d, int S, int head_mask, int fusion_mode, int I_total) -> Tensor");
    m.def("score_topk_shard(Tensor Y, int U, int I, Tensor users, int d, int S, int head_mask, int fusion_mode, int predict_type, Tensor? train_ptr, Tensor? train_items, int K, Tensor row_sum, int I_total, int id_offset) -> (Tensor, Tensor)");
    m.def("topk_merge(Tensor cand_val, Tensor cand_idx, int K) -> (Tensor, Tensor)");
    m.def("lookup_counts(Tensor acts, int U, int I, int[] user_bounds, int[] item_bounds) -> Tensor");
    m.def("lookup_pack(Tensor acts, int U, int I, int[] user_bounds, int[] item_bounds, int me, Tensor shard, int row_bytes) -> (Tensor, Tensor)");
    m.def("lookup_unpack(Tensor act, int U, int I, int[] user_bounds, int[] item_bounds, int me, Tensor rows, int row_bytes, int dtype, int sum_d, bool direct) -> (Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(elimrec, CUDA, m) {
    m.impl("propagate", &propagate);
    m.impl("segment_reduce", &segment_reduce);
    m.impl("linear_fwd", &linear_fwd);
    m.impl("linear_bwd_w", &linear_bwd_w);
    m.impl("bpr_head_fwd", &bpr_head_fwd);
    m.impl("adam_step_", &adam_step_);
    m.impl("score_topk", &score_topk);
    m.impl("rank_metrics", &rank_metrics);
    m.impl("sample_triplets", &sample_triplets);
    m.impl("bpr_head_bwd", &bpr_head_bwd);
    m.impl("score_shard_row_sums", &score_shard_row_sums);
    m.impl("score_topk_shard", &score_topk_shard);
    m.impl("topk_merge", &topk_merge);
    m.impl("lookup_counts", &lookup_counts);
    m.impl("lookup_pack", &lookup_pack);
    m.impl("lookup_unpack", &lookup_unpack);
}
