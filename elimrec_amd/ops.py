"""Tensor-level wrappers over the C ABI (include/elimrec_hip.h).

torch is plumbing here: it owns device memory and the current HIP stream; every function below
hands raw device pointers to libelimrec_hip.so. Inputs must live on a HIP device -- anything
else raises (no CPU path).
"""
import ctypes

import torch

from . import _lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Handle of torch's current HIP stream on the current device (the fast private accessor when present)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _dev(t, name, dtype=torch.float32):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("elimrec_amd.ops: '%s' must be a HIP device tensor (the hot path has no CPU "
                           "implementation)" % name)
    if t.dtype != dtype:
        raise TypeError("elimrec_amd.ops: '%s' must be %s, got %s" % (name, dtype, t.dtype))
    return t.data_ptr()


def _rowmajor(t, name):
    """(pointer, leading dimension) of a 2-D tensor whose rows are contiguous."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("elimrec_amd.ops: '%s' must be 2-D with unit column stride" % name)
    return _dev(t, name), t.stride(0)


def linear_fwd(A, W, bias, out, act=None):
    """out[m, n] = act(sum_k A[m,k] W[n,k] + bias[n]); A/out may be column slices of wider tables. act: None | 'relu'."""
    lib = _lib.load()
    a, lda = _rowmajor(A, "A")
    w, ldw = _rowmajor(W, "W")
    c, ldc = _rowmajor(out, "out")
    M, K = A.shape
    N = W.shape[0]
    assert W.shape[1] == K and out.shape[0] == M and out.shape[1] == N
    if act is None:
        _lib.check(lib.elimrec_linear_fwd(a, lda, w, ldw, _dev(bias, "bias"), c, ldc, M, N, K, _stream()), "linear_fwd")
        return out
    if act != "relu":
        raise ValueError("linear_fwd: the fused epilogue knows 'relu' only (got %r)" % (act,))
    arr = (_lib.LinearDesc * 1)(_lib.LinearDesc(a, lda, w, ldw, _dev(bias, "bias"), c, ldc, M, N, K, None, None, 0, None, None, 1))
    _lib.check(lib.elimrec_linear_fwd_batched(arr, 1, _stream()), "linear_fwd(relu)")
    return out


def linear_fwd_batched(problems):
    """problems: list of (A, W, bias, out[, rowscale, add[, row_index, row_count]]) -- independent Linears sharing
    one launch (<= 8).  out = A.W^T + rowscale[:, None] * bias + add   (rowscale, add optional).
    row_index (int32 [M]): output row m reads row row_index[m] of A / rowscale / add; row_range (device int32[2]):
    only output rows [begin, min(end, M)) are produced."""
    n = len(problems)
    arr = (_lib.LinearDesc * n)()
    for i, pr in enumerate(problems):
        A, W, bias, out = pr[:4]
        rowscale = pr[4] if len(pr) > 4 else None
        add = pr[5] if len(pr) > 5 else None
        row_index = pr[6] if len(pr) > 6 else None
        row_range = pr[7] if len(pr) > 7 else None
        a, lda = _rowmajor(A, "A")
        w, ldw = _rowmajor(W, "W")
        c, ldc = _rowmajor(out, "out")
        K = A.shape[1]
        M = A.shape[0] if row_index is None else row_index.numel()
        N = W.shape[0]
        assert W.shape[1] == K and out.shape[0] == M and out.shape[1] == N
        ad, ldadd = (None, 0)
        if add is not None:
            assert add.shape == (A.shape[0], N)
            ad, ldadd = _rowmajor(add, "add")
        if rowscale is not None:
            assert rowscale.is_contiguous() and rowscale.numel() == A.shape[0]
        arr[i] = _lib.LinearDesc(a, lda, w, ldw, _dev(bias, "bias"), c, ldc, M, N, K, _dev(rowscale, "rowscale"), ad, ldadd,
                                 _dev(row_index, "row_index", torch.int32), _dev(row_range, "row_range", torch.int32))
    _lib.check(_lib.load().elimrec_linear_fwd_batched(arr, n, _stream()), "linear_fwd_batched")


def _bwd_descs(problems):
    n = len(problems)
    arr = (_lib.LinearBwdDesc * n)()
    for i, pr in enumerate(problems):
        a, lda = _rowmajor(pr["A"], "A")
        b, ldb = _rowmajor(pr["B"], "B")
        o, ldo = _rowmajor(pr["out"], "out")
        R = pr["A"].shape[0] if pr.get("rows") is None else pr["rows"]
        n1, n2 = pr["out"].shape
        arr[i] = _lib.LinearBwdDesc(a, lda, b, ldb, _dev(pr.get("row_index"), "row_index", torch.int32),
                                    _dev(pr.get("rng"), "range", torch.int32), R, n1, n2, o, ldo,
                                    _dev(pr.get("colsum"), "colsum"), 1 if pr.get("accumulate") else 0,
                                    _dev(pr.get("colsum_weight"), "colsum_weight"))
    return arr, n


def linear_bwd_w_batched_workspace(shapes):
    """shapes: list of (R, n1, n2)."""
    n = len(shapes)
    arr = (_lib.LinearBwdDesc * n)()
    for i, (R, n1, n2) in enumerate(shapes):
        arr[i] = _lib.LinearBwdDesc(None, 0, None, 0, None, None, R, n1, n2, None, 0, None, 0)
    return int(_lib.load().elimrec_linear_bwd_w_batched_workspace(arr, n))


def linear_bwd_w_batched(problems, workspace, merge=None, defer_reduce=False, defer_all=False):
    """problems: list of dicts(A, B, out[, row_index, rng, colsum, accumulate, rows]) in one launch pair.
    merge: dict(rows, keys, world, U, I, srcA, srcB, mask, M) -- the arguments of slab.merge_rows, run as extra workgroups
    of the partial launch (elimrec_linear_bwd_w_batched_merge).
    defer_reduce: stop after the partial launch and return the handle `linear_bwd_w_reduce` / `slab.hop(bwd_w=)` finish
    the gradients with (the outputs hold nothing until then). defer_all: launch nothing, return the handle: both phases
    ride in hop launches (slab.hop(bwd_w=handle, bwd_w_phase=0), then bwd_w_phase=1)."""
    arr, n = _bwd_descs(problems)
    if defer_all:
        assert merge is None
        need = int(_lib.load().elimrec_linear_bwd_w_batched_workspace(arr, n))
        assert workspace.numel() >= need, "linear_bwd_w: workspace too small"
        return (arr, n, workspace)
    wsp, wsn = _dev(workspace, "workspace", torch.uint8), workspace.numel()
    if merge is None and not defer_reduce:
        _lib.check(_lib.load().elimrec_linear_bwd_w_batched(arr, n, wsp, wsn, _stream()), "linear_bwd_w_batched")
        return None
    if merge is not None:
        rows, keys, srcA, srcB, M, world = merge["rows"], merge["keys"], merge["srcA"], merge["srcB"], merge["M"], merge["world"]
        R = keys.numel() // world
        assert rows.is_contiguous() and rows.shape == (world * R, (M if M else 2) * srcA.cols)
        assert merge["mask"].numel() * 32 >= merge["U"] + merge["I"]
        margs = (_dev(rows, "rows"), _dev(keys, "keys", torch.int32), int(world), R, int(merge["U"]), int(merge["I"]), srcA.ns, srcA.w,
                 int(M), _dev(srcA.data, "srcA"), _dev(srcB.data, "srcB"), _dev(merge["mask"], "mask", torch.int32))
    else:
        margs = (None, None, 1, 1, 0, 0, 1, 4, 0, None, None, None)
    _lib.check(_lib.load().elimrec_linear_bwd_w_batched_merge(arr, n, wsp, wsn, *margs, 1 if defer_reduce else 0, _stream()),
               "linear_bwd_w_batched_merge")
    return (arr, n, workspace) if defer_reduce else None


def linear_bwd_w_reduce(handle):
    """The fixed-order slab reduce of a `linear_bwd_w_batched(..., defer_reduce=True)` launch, as a launch of its own."""
    arr, n, workspace = handle
    _lib.check(_lib.load().elimrec_linear_bwd_w_reduce(arr, n, _dev(workspace, "workspace", torch.uint8), workspace.numel(), _stream()),
               "linear_bwd_w_reduce")


def linear_bwd_w_workspace(R, n1, n2):
    return int(_lib.load().elimrec_linear_bwd_w_workspace(R, n1, n2))


def linear_bwd_w(A, B, out, workspace, row_index=None, rng=None, colsum=None, accumulate=False, rows=None):
    """out[i, j] (+)= sum_r A[r, i] * B[row_index[r] or r, j]; colsum[i] (+)= sum_r A[r, i]."""
    lib = _lib.load()
    a, lda = _rowmajor(A, "A")
    b, ldb = _rowmajor(B, "B")
    o, ldo = _rowmajor(out, "out")
    R = A.shape[0] if rows is None else rows
    n1, n2 = out.shape
    _lib.check(lib.elimrec_linear_bwd_w(a, lda, b, ldb, _dev(row_index, "row_index", torch.int32),
                                        _dev(rng, "range", torch.int32), R, n1, n2, o, ldo, _dev(colsum, "colsum"),
                                        1 if accumulate else 0, _dev(workspace, "workspace", torch.uint8),
                                        workspace.numel(), _stream()), "linear_bwd_w")
    return out


def assemble_x0(user_emb, item_emb, X0, M):
    U, d = user_emb.shape
    I = item_emb.shape[0]
    assert X0.is_contiguous() and X0.shape == (U + I, M * d)
    assert user_emb.is_contiguous() and item_emb.is_contiguous()
    _lib.check(_lib.load().elimrec_assemble_x0(_dev(user_emb, "user_emb"), _dev(item_emb, "item_emb"), _dev(X0, "X0"),
                                               U, I, d, M, _stream()), "assemble_x0")
    return X0


LONG_ROW_THRESHOLD = 64   # rows with more non-zeros are split across waves (csrc/spmm.hip)


class Csr:
    """Device CSR (int32 rowptr/col, fp32 val) + the optional row-split plan for long rows."""

    def __init__(self, rowptr, col, val, n_rows):
        self.rowptr, self.col, self.val, self.n_rows = rowptr, col, val, n_rows
        self._split = None
        self._split_tensors = None

    @staticmethod
    def from_scipy(m, device, C=None, threshold=LONG_ROW_THRESHOLD):
        import numpy as np
        m = m.tocsr()
        m.sort_indices()
        if m.nnz >= 2 ** 31:
            raise ValueError("CSR with >= 2^31 non-zeros is not supported")
        csr = Csr(torch.from_numpy(m.indptr.astype(np.int32)).to(device),
                  torch.from_numpy(m.indices.astype(np.int32)).to(device),
                  torch.from_numpy(m.data.astype(np.float32)).to(device), m.shape[0])
        if C is not None:
            csr.build_split(C, threshold)
        return csr

    def build_split(self, C, threshold=LONG_ROW_THRESHOLD):
        """Cut rows with more than `threshold` non-zeros into segments of <= threshold (host, once)."""
        import numpy as np
        rowptr = self.rowptr.cpu().numpy().astype(np.int64)
        deg = np.diff(rowptr)
        long_rows = np.nonzero(deg > threshold)[0]
        dev = self.rowptr.device
        # rows in order of decreasing length (split rows count as empty in the main pass): a scheduling hint
        order = np.argsort(-np.where(deg > threshold, 0, deg), kind="stable").astype(np.int32)
        self._row_order = torch.from_numpy(order).to(dev)
        ro = self._row_order.data_ptr()
        # (row, begin, end) of the rows that are not split, in that order: the streaming form of the row kernel
        keep = order[deg[order] <= threshold].astype(np.int64)
        items = np.stack([keep, rowptr[keep], rowptr[keep + 1]], 1).astype(np.int32)
        self._row_items = torch.from_numpy(np.ascontiguousarray(items)).to(dev)
        ri, n_items = self._row_items.data_ptr(), int(len(keep))
        if len(long_rows) == 0:
            self._split_tensors = None
            self._split = _lib.CsrSplit(0, 0, 0, None, None, None, None, None, ro, None, ri, n_items)
            self._split_C = C
            return self
        nseg = (deg[long_rows] + threshold - 1) // threshold
        seg_ptr = np.concatenate([[0], np.cumsum(nseg)])
        total = int(seg_ptr[-1])
        seg_row = np.repeat(np.arange(len(long_rows)), nseg)
        k = np.arange(total) - seg_ptr[seg_row]
        beg = rowptr[long_rows][seg_row] + k * threshold
        end = np.minimum(beg + threshold, rowptr[long_rows + 1][seg_row])
        t = (torch.from_numpy(long_rows.astype(np.int32)).to(dev), torch.from_numpy(seg_ptr.astype(np.int32)).to(dev),
             torch.from_numpy(np.stack([beg, end], 1).astype(np.int32).copy()).to(dev),
             torch.empty(total, 2 * C, dtype=torch.float32, device=dev),   # wide + narrow partial regions
             torch.from_numpy(seg_row.astype(np.int32)).to(dev),
             torch.zeros(2 * len(long_rows), dtype=torch.int32, device=dev))
        self._split_tensors = t
        self._split = _lib.CsrSplit(int(threshold), len(long_rows), total, t[0].data_ptr(), t[1].data_ptr(),
                                    t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr(), ro, t[5].data_ptr(), ri, n_items)
        self._split_C = C
        return self

    def desc(self):
        """struct elimrec_csr for the block-CSR entry points (keeps the tensors alive through self)."""
        sp = self._split if self._split is not None else _lib.CsrSplit(0, 0, 0, None, None, None, None, None, None, None, None, 0)
        self._desc = _lib.CsrDesc(self.n_rows, self.rowptr.data_ptr(), self.col.data_ptr(), self.val.data_ptr(), sp)
        return ctypes.byref(self._desc)

    def split_ref(self, C):
        if self._split is None or self._split.n_long == 0:
            return None
        if self._split_C < C:
            raise ValueError("row-split plan was built for C=%d, got C=%d" % (self._split_C, C))
        return ctypes.byref(self._split)


def spmm_hop(csr, Xin, Xout=None, acc_in=None, acc_out=None, scale=1.0):
    C = Xin.shape[1]
    assert Xin.is_contiguous()
    _lib.check(_lib.load().elimrec_spmm_hop(_dev(csr.rowptr, "rowptr", torch.int32), _dev(csr.col, "col", torch.int32),
                                            _dev(csr.val, "val"), csr.n_rows, C, csr.split_ref(C), _dev(Xin, "Xin"),
                                            _dev(Xout, "Xout"), _dev(acc_in, "acc_in"), _dev(acc_out, "acc_out"),
                                            float(scale), _stream()), "spmm_hop")


def propagate(csr, X0, L, tmp0, tmp1, out):
    C = X0.shape[1]
    assert X0.is_contiguous() and out.is_contiguous() and out.shape == X0.shape
    _lib.check(_lib.load().elimrec_propagate(_dev(csr.rowptr, "rowptr", torch.int32), _dev(csr.col, "col", torch.int32),
                                             _dev(csr.val, "val"), csr.n_rows, C, csr.split_ref(C), L, _dev(X0, "X0"),
                                             _dev(tmp0, "tmp0"), _dev(tmp1, "tmp1"), _dev(out, "out"), _stream()),
               "propagate")
    return out


def bipartite_workspace(U, I, d, M):
    return int(_lib.load().elimrec_bipartite_workspace(U, I, d, M))


def propagate_bipartite(P, Q, U, I, d, M, L, user_emb, XI, out, workspace, narrow_out=None):
    assert user_emb.is_contiguous() and XI.is_contiguous() and out.is_contiguous()
    assert XI.shape == (I, d * M) and out.shape == (U + I, d * M) and user_emb.shape == (U, d)
    assert narrow_out is None or (narrow_out.is_contiguous() and narrow_out.shape == (U + I, d))
    _lib.check(_lib.load().elimrec_propagate_bipartite(P.desc(), Q.desc(), U, I, d, M, L, _dev(user_emb, "user_emb"),
                                                       _dev(XI, "XI"), _dev(out, "out"), _dev(narrow_out, "narrow_out"),
                                                       _dev(workspace, "workspace", torch.uint8), workspace.numel(),
                                                       _stream()), "propagate_bipartite")
    return out


def propagate_bipartite_bwd(PT, QT, U, I, d, M, L, G, H, active_rows, seg_info, gXI, gEu, workspace):
    assert G.is_contiguous() and H.is_contiguous() and gXI.is_contiguous() and gEu.is_contiguous()
    assert G.shape == (U + I, d * M) and H.shape == (U + I, d) and gXI.shape == (I, d * M) and gEu.shape == (U, d)
    _lib.check(_lib.load().elimrec_propagate_bipartite_bwd(PT.desc(), QT.desc(), U, I, d, M, L, _dev(G, "G"), _dev(H, "H"),
                                                           _dev(active_rows, "active_rows", torch.int32),
                                                           _dev(seg_info, "seg_info", torch.int32), active_rows.numel(),
                                                           _dev(gXI, "gXI"), _dev(gEu, "gEu"),
                                                           _dev(workspace, "workspace", torch.uint8), workspace.numel(),
                                                           _stream()), "propagate_bipartite_bwd")


def folded_workspace(N, d):
    return int(_lib.load().elimrec_folded_workspace(N, d))


def propagate_folded(A, U, I, d, L, X0, out0, narrow, workspace):
    """out0: [N x d] window (unit column stride) of a wider table; X0, narrow: contiguous [N x d]."""
    assert X0.is_contiguous() and narrow.is_contiguous() and X0.shape == (U + I, d) and narrow.shape == (U + I, d)
    o, ldo = _rowmajor(out0, "out0")
    assert out0.shape == (U + I, d)
    _lib.check(_lib.load().elimrec_propagate_folded(A.desc(), U, I, d, L, _dev(X0, "X0"), o, ldo, _dev(narrow, "narrow"),
                                                    _dev(workspace, "workspace", torch.uint8), workspace.numel(),
                                                    _stream()), "propagate_folded")


def source_rows_split(dOutR, count, d, M, world, out):
    """out[w][s] = [H | G] of dOutR[s] (H = sum of its M column blocks, G = block 0) in peer w's column slice, s < count:
    out [world x n x 2*d/world]."""
    n = dOutR.shape[0]
    assert dOutR.is_contiguous() and out.is_contiguous() and dOutR.shape[1] == d * M and out.shape == (world, n, 2 * d // world)
    _lib.check(_lib.load().elimrec_source_rows_split(_dev(dOutR, "dOutR"), _dev(count, "count", torch.int32), n, d, M, world,
                                                     _dev(out, "out"), _stream()), "source_rows_split")


def propagate_folded_bwd(AT, U, I, d, M, L, dOutR, active_rows, seg_info, srcA, srcB, grad, workspace, active_mask=None):
    """dOutR None: srcA / srcB and active_mask were prefilled by the caller."""
    for t in (srcA, srcB, grad):
        assert t.is_contiguous()
    assert grad.shape == (U + I, d) and (dOutR is None or (dOutR.is_contiguous() and dOutR.shape[1] == d * M))
    _lib.check(_lib.load().elimrec_propagate_folded_bwd(AT.desc(), U, I, d, M, L, _dev(dOutR, "dOutR"),
                                                        _dev(active_rows, "active_rows", torch.int32),
                                                        _dev(seg_info, "seg_info", torch.int32),
                                                        0 if active_rows is None else active_rows.numel(),
                                                        _dev(srcA, "srcA"), _dev(srcB, "srcB"), _dev(grad, "grad"),
                                                        _dev(active_mask, "active_mask", torch.int32),
                                                        _dev(workspace, "workspace", torch.uint8), workspace.numel(),
                                                        _stream()), "propagate_folded_bwd")


def block_spmm(A, Xin, Xout=None, add1=None, acc_out=None, scale=1.0):
    """r = A . Xin (a column window of a wider table is fine); Xout = r; acc_out = (r + add1) * scale."""
    x, ld = _rowmajor(Xin, "Xin")
    W = Xin.shape[1]
    for t in (Xout, add1, acc_out):
        assert t is None or (t.stride(0) == ld and t.stride(1) == 1 and t.shape[1] == W)
    _lib.check(_lib.load().elimrec_block_spmm(A.desc(), W, ld, x, _dev(Xout, "Xout"), _dev(add1, "add1"),
                                              _dev(acc_out, "acc_out"), float(scale), _stream()), "block_spmm")


def blocksum_rows(G, active_rows, seg_info, d, M, H, slot_major=False):
    """H[node] = sum over the M column blocks of the node's row of G (rows of G indexed by node, or by slot)."""
    assert G.is_contiguous() and H.is_contiguous()
    _lib.check(_lib.load().elimrec_blocksum_rows(_dev(G, "G"), _dev(active_rows, "active_rows", torch.int32),
                                                 _dev(seg_info, "seg_info", torch.int32), active_rows.numel(), d, M,
                                                 1 if slot_major else 0, _dev(H, "H"), _stream()), "blocksum_rows")


def copy_cols(src, dst):
    s, lds = _rowmajor(src, "src")
    t, ldt = _rowmajor(dst, "dst")
    assert src.shape == dst.shape
    _lib.check(_lib.load().elimrec_copy_cols(s, lds, t, ldt, src.shape[0], src.shape[1], _stream()), "copy_cols")
    return dst


def triplet_rows(users, pos, neg, U, rows, src=None, dst=None, I=None, err=None):
    """rows[3b + (0,1,2)] = users[b], U + pos[b], U + neg[b]; dst[slot] = src[rows[slot]] when src is given. With I and
    err (int32[1]): range-checked -- a bad index sets a bit of err and is replaced by 0."""
    B = users.numel()
    if err is not None:
        _lib.check(_lib.load().elimrec_triplet_rows_checked(_dev(users, "users", torch.int64), _dev(pos, "pos", torch.int64),
                                                            _dev(neg, "neg", torch.int64), B, U, int(I),
                                                            _dev(rows, "rows", torch.int32), _dev(err, "err", torch.int32),
                                                            _stream()), "triplet_rows_checked")
        return rows
    s, lds, t, ldt, cols = None, 0, None, 0, 0
    if src is not None:
        s, lds = _rowmajor(src, "src")
        t, ldt = _rowmajor(dst, "dst")
        cols = src.shape[1]
        assert dst.shape == (3 * B, cols)
    _lib.check(_lib.load().elimrec_triplet_rows(_dev(users, "users", torch.int64), _dev(pos, "pos", torch.int64),
                                                _dev(neg, "neg", torch.int64), B, U, _dev(rows, "rows", torch.int32),
                                                s, lds, cols, t, ldt, _stream()), "triplet_rows")
    return rows


def batch_plan(users, pos, neg, U, I, keys, active_rows, seg_info, slot_seg, workspace, err, pad_key, key_bitmap=None):
    """triplet_rows (range-checked) + segment_plan + padding of the active-row list in one launch (elimrec_batch_plan)."""
    B = users.numel()
    _lib.check(_lib.load().elimrec_batch_plan(_dev(users, "users", torch.int64), _dev(pos, "pos", torch.int64),
                                              _dev(neg, "neg", torch.int64), B, int(U), int(I), _dev(keys, "keys", torch.int32),
                                              _dev(active_rows, "active_rows", torch.int32),
                                              _dev(seg_info, "seg_info", torch.int32), _dev(slot_seg, "slot_seg", torch.int32),
                                              _dev(key_bitmap, "key_bitmap", torch.int32), int(pad_key),
                                              _dev(err, "err", torch.int32), _dev(workspace, "workspace", torch.uint8),
                                              workspace.numel(), _stream()), "batch_plan")
    return keys


def gather_rows(src, rows, dst, count=None):
    """dst[r] = src[rows[r]] for r < min(count, len(rows))."""
    s, lds = _rowmajor(src, "src")
    t, ldt = _rowmajor(dst, "dst")
    assert dst.shape == (rows.numel(), src.shape[1])
    _lib.check(_lib.load().elimrec_gather_rows(s, lds, _dev(rows, "rows", torch.int32), _dev(count, "count", torch.int32),
                                               rows.numel(), src.shape[1], t, ldt, _stream()), "gather_rows")
    return dst


def bpr_head(Y, U, I, users, pos, neg, d, block_weights, loss_rows, grad_rows=None, keys=None):
    y, ldy = _rowmajor(Y, "Y")
    nb = len(block_weights)
    w = (ctypes.c_float * nb)(*[float(x) for x in block_weights])
    B = users.numel()
    _lib.check(_lib.load().elimrec_bpr_head(y, ldy, U, I, _dev(users, "users", torch.int64), _dev(pos, "pos", torch.int64),
                                            _dev(neg, "neg", torch.int64), B, d, nb, w, _dev(loss_rows, "loss_rows"),
                                            _dev(grad_rows, "grad_rows"), _dev(keys, "keys", torch.int32), _stream()),
               "bpr_head")


def bpr_head_rows(Y, slot_rows, d, block_weights, loss_rows, grad_rows=None):
    """bpr_head over a compact table: slot 3b+j of triplet b reads row slot_rows[3b+j] of Y."""
    y, ldy = _rowmajor(Y, "Y")
    nb = len(block_weights)
    w = (ctypes.c_float * nb)(*[float(x) for x in block_weights])
    B = slot_rows.numel() // 3
    _lib.check(_lib.load().elimrec_bpr_head_rows(y, ldy, _dev(slot_rows, "slot_rows", torch.int32), B, d, nb, w,
                                                 _dev(loss_rows, "loss_rows"), _dev(grad_rows, "grad_rows"), _stream()),
               "bpr_head_rows")


def bpr_head_rows_sum(Y, slot_rows, d, block_weights, loss_rows, grad_rows, loss_out, ticket):
    """bpr_head_rows + the fixed-order sum of its loss rows into loss_out (0-dim / 1-element fp32) in one launch; ticket: a
    zero-initialised int32 the kernel leaves at zero."""
    y, ldy = _rowmajor(Y, "Y")
    nb = len(block_weights)
    w = (ctypes.c_float * nb)(*[float(x) for x in block_weights])
    B = slot_rows.numel() // 3
    _lib.check(_lib.load().elimrec_bpr_head_rows_sum(y, ldy, _dev(slot_rows, "slot_rows", torch.int32), B, d, nb, w,
                                                     _dev(loss_rows, "loss_rows"), _dev(grad_rows, "grad_rows"),
                                                     _dev(loss_out, "loss_out"), _dev(ticket, "ticket", torch.int32), _stream()),
               "bpr_head_rows_sum")


def bpr_head_rows_sum_pub(Y, slot_rows, d, block_weights, loss_rows, grad_rows, loss_out, ticket, pub):
    """bpr_head_rows_sum whose summing workgroup also publishes the loss to the host (pub: LossPublisher.handle)."""
    y, ldy = _rowmajor(Y, "Y")
    nb = len(block_weights)
    w = (ctypes.c_float * nb)(*[float(x) for x in block_weights])
    B = slot_rows.numel() // 3
    _lib.check(_lib.load().elimrec_bpr_head_rows_sum_pub(y, ldy, _dev(slot_rows, "slot_rows", torch.int32), B, d, nb, w,
                                                         _dev(loss_rows, "loss_rows"), _dev(grad_rows, "grad_rows"),
                                                         _dev(loss_out, "loss_out"), _dev(ticket, "ticket", torch.int32), pub, _stream()),
               "bpr_head_rows_sum_pub")


class LossPublisher(object):
    """elimrec_loss_pub_*: a ring of coherent host words the loss-summing launch of a step writes (sequence number, value) into,
    so that the caller's `loss.item()` (main.py:102 of the reference) waits for THAT launch and not for the whole step."""

    def __init__(self, n_slots=64):
        h = ctypes.c_void_p()
        _lib.check(_lib.load().elimrec_loss_pub_create(int(n_slots), ctypes.byref(h)), "loss_pub_create")
        self.handle = h
        self._val = ctypes.c_float()

    def issued(self):
        return int(_lib.load().elimrec_loss_pub_issued(self.handle))

    def wait(self, seq, timeout_s=60.0):
        """The loss launch `seq` published, or None when the ring has wrapped past it."""
        rc = _lib.load().elimrec_loss_pub_wait(self.handle, int(seq), float(timeout_s), ctypes.byref(self._val))
        if rc == 0:
            return float(self._val.value)
        if rc == 10002:            # ELIMREC_E_UNSUPPORTED: overwritten by a later step
            return None
        _lib.check(rc, "loss_pub_wait")

    def __del__(self):
        try:
            if self.handle:
                _lib.load().elimrec_loss_pub_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def fixed_order_sum(x, out):
    _lib.check(_lib.load().elimrec_sum(_dev(x, "x"), x.numel(), _dev(out, "out"), _stream()), "sum")
    return out


def segment_reduce_workspace(n):
    return int(_lib.load().elimrec_segment_reduce_workspace(n))


def segment_plan_workspace(n):
    return int(_lib.load().elimrec_segment_plan_workspace(n))


def segment_plan(keys, split_key, key_space, active_rows, seg_info, slot_seg, workspace, key_bitmap=None):
    """Plan of a key list: active_rows (sorted unique keys), seg_info, slot_seg (segment of every slot); the member
    lists stay in `workspace` for segment_apply. key_bitmap (int32 words, >= key_space bits): bit k <=> k is active."""
    n = keys.numel()
    assert key_bitmap is None or key_bitmap.numel() * 32 >= key_space
    _lib.check(_lib.load().elimrec_segment_plan(_dev(keys, "keys", torch.int32), n, int(split_key), int(key_space),
                                                _dev(active_rows, "active_rows", torch.int32),
                                                _dev(seg_info, "seg_info", torch.int32),
                                                _dev(slot_seg, "slot_seg", torch.int32),
                                                _dev(key_bitmap, "key_bitmap", torch.int32),
                                                _dev(workspace, "workspace", torch.uint8), workspace.numel(), _stream()),
               "segment_plan")


def segment_apply(rows, seg_info, reduced, workspace, scale=None):
    n, ld = rows.shape
    assert rows.is_contiguous() and reduced.is_contiguous()
    _lib.check(_lib.load().elimrec_segment_apply(_dev(rows, "rows"), n, ld, _dev(seg_info, "seg_info", torch.int32),
                                                 _dev(scale, "scale"), _dev(reduced, "reduced"),
                                                 _dev(workspace, "workspace", torch.uint8), workspace.numel(), _stream()),
               "segment_apply")


def segment_reduce_rows(rows, keys, split_key, active_rows, reduced, seg_info, workspace, scale=None):
    n, ld = rows.shape
    assert rows.is_contiguous() and reduced.is_contiguous()
    _lib.check(_lib.load().elimrec_segment_reduce_rows(_dev(rows, "rows"), _dev(keys, "keys", torch.int32), n, ld,
                                                       int(split_key), _dev(active_rows, "active_rows", torch.int32),
                                                       _dev(reduced, "reduced"), _dev(scale, "scale"),
                                                       _dev(seg_info, "seg_info", torch.int32),
                                                       _dev(workspace, "workspace", torch.uint8), workspace.numel(),
                                                       _stream()), "segment_reduce_rows")


def head_bwd_input(dY, active_rows, seg_info, U, d, C, head_mblock, W_user, W_item, W_heads, gscale, G0,
                   scatter_cols=None, compact=None):
    """G0[node, 0:scatter_cols] (row stride G0.stride(0)) and/or compact[slot, 0:C] receive the input gradient."""
    S = len(W_heads)
    dy, lddy = _rowmajor(dY, "dY")
    mb = (ctypes.c_int * max(S, 1))(*head_mblock) if S else (ctypes.c_int * 1)(0)
    wp = (ctypes.c_void_p * max(S, 1))(*[_dev(w, "W_head") for w in W_heads]) if S else (ctypes.c_void_p * 1)(None)
    for w in list(W_heads) + [W_user, W_item]:
        assert w.is_contiguous()
    g0, ldg = (None, 0)
    if G0 is not None:
        g0, ldg = _rowmajor(G0, "G0")
    if scatter_cols is None:
        scatter_cols = C if G0 is not None else 0
    assert compact is None or (compact.is_contiguous() and compact.shape[1] == C and compact.shape[0] >= dY.shape[0])
    _lib.check(_lib.load().elimrec_head_bwd_input(dy, lddy, _dev(active_rows, "active_rows", torch.int32),
                                                  _dev(seg_info, "seg_info", torch.int32), dY.shape[0], U, d, C, S, mb,
                                                  _dev(W_user, "W_user"), _dev(W_item, "W_item"), wp, float(gscale),
                                                  g0, ldg, int(scatter_cols), _dev(compact, "compact"), _stream()),
               "head_bwd_input")


def segment_apply_head_bwd(rows, active_rows, seg_info, reduced, plan_workspace, U, d, C, head_mblock, W_user, W_item,
                           W_heads, compact, scale=None, pack_bwd=None, sources=None):
    """segment_apply + head_bwd_input(compact=...) in one launch; `reduced` receives dY. pack_bwd: the backward region of
    the fused head's packed weights (a view starting at head_pack_bwd_offset floats), if the forward left it behind.
    sources = (srcA, srcB) slab tables: the kernel also fills the adjoint sources at the active rows (one rank, recdim 64,
    packed weights: elimrec_segment_apply_head_bwd_sources); ("split", send, world): the peers' [H | G] column slices."""
    n, ld = rows.shape
    S = len(W_heads)
    assert rows.is_contiguous() and reduced.is_contiguous() and reduced.shape[1] == ld and compact.is_contiguous()
    assert compact.shape[1] == C and compact.shape[0] >= n and reduced.shape[0] >= n
    mb = (ctypes.c_int * max(S, 1))(*head_mblock) if S else (ctypes.c_int * 1)(0)
    wp = (ctypes.c_void_p * max(S, 1))(*[_dev(w, "W_head") for w in W_heads]) if S else (ctypes.c_void_p * 1)(None)
    for w in list(W_heads) + [W_user, W_item]:
        assert w.is_contiguous()
    if sources is not None and sources[0] == "split":          # ("split", send [W x n_max x 2*dl], W)
        _, send, world = sources
        assert pack_bwd is not None and send.is_contiguous() and send.shape == (world, send.shape[1], 2 * (d // world)) and send.shape[1] >= n
        _lib.check(_lib.load().elimrec_segment_apply_head_bwd_split(
            _dev(rows, "rows"), n, ld, _dev(active_rows, "active_rows", torch.int32), _dev(seg_info, "seg_info", torch.int32),
            _dev(scale, "scale"), _dev(reduced, "reduced"), _dev(plan_workspace, "plan_workspace", torch.uint8),
            plan_workspace.numel(), U, d, C, S, mb, _dev(W_user, "W_user"), _dev(W_item, "W_item"), wp,
            _dev(compact, "compact"), _dev(pack_bwd, "pack_bwd"), send.shape[1], int(world), _dev(send, "send"), _stream()),
            "segment_apply_head_bwd_split")
        return
    if sources is not None:
        srcA, srcB = sources
        assert pack_bwd is not None and srcA.ns == srcB.ns and srcA.w == srcB.w and srcA.n == srcB.n
        _lib.check(_lib.load().elimrec_segment_apply_head_bwd_sources(
            _dev(rows, "rows"), n, ld, _dev(active_rows, "active_rows", torch.int32), _dev(seg_info, "seg_info", torch.int32),
            _dev(scale, "scale"), _dev(reduced, "reduced"), _dev(plan_workspace, "plan_workspace", torch.uint8),
            plan_workspace.numel(), U, d, C, S, mb, _dev(W_user, "W_user"), _dev(W_item, "W_item"), wp,
            _dev(compact, "compact"), _dev(pack_bwd, "pack_bwd"), srcA.n, srcA.ns, srcA.w, _dev(srcA.data, "srcA"),
            _dev(srcB.data, "srcB"), _stream()), "segment_apply_head_bwd_sources")
        return
    if pack_bwd is not None:
        _lib.check(_lib.load().elimrec_segment_apply_head_bwd_packed(
            _dev(rows, "rows"), n, ld, _dev(active_rows, "active_rows", torch.int32), _dev(seg_info, "seg_info", torch.int32),
            _dev(scale, "scale"), _dev(reduced, "reduced"), _dev(plan_workspace, "plan_workspace", torch.uint8),
            plan_workspace.numel(), U, d, C, S, mb, _dev(W_user, "W_user"), _dev(W_item, "W_item"), wp,
            _dev(compact, "compact"), _dev(pack_bwd, "pack_bwd"), _stream()), "segment_apply_head_bwd_packed")
        return
    _lib.check(_lib.load().elimrec_segment_apply_head_bwd(
        _dev(rows, "rows"), n, ld, _dev(active_rows, "active_rows", torch.int32), _dev(seg_info, "seg_info", torch.int32),
        _dev(scale, "scale"), _dev(reduced, "reduced"), _dev(plan_workspace, "plan_workspace", torch.uint8),
        plan_workspace.numel(), U, d, C, S, mb, _dev(W_user, "W_user"), _dev(W_item, "W_item"), wp,
        _dev(compact, "compact"), _stream()), "segment_apply_head_bwd")


def embed_grad(G, U, I, d, M, grad_user, grad_item):
    assert G.is_contiguous() and grad_user.is_contiguous() and grad_item.is_contiguous()
    _lib.check(_lib.load().elimrec_embed_grad(_dev(G, "G"), U, I, d, M, _dev(grad_user, "grad_user"),
                                              _dev(grad_item, "grad_item"), _stream()), "embed_grad")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step):
    for t in (p, g, m, v):
        assert t.is_contiguous()
    _lib.check(_lib.load().elimrec_adam_step(_dev(p, "p"), _dev(g, "g"), _dev(m, "m"), _dev(v, "v"), p.numel(),
                                             float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                             int(step), _stream()), "adam_step")


def adam_step_raw(p_ptr, g_ptr, m_ptr, v_ptr, n, lr, beta1, beta2, eps, weight_decay, step):
    """Same kernel on raw device addresses (a span covering several adjacent tensors)."""
    _lib.check(_lib.load().elimrec_adam_step(p_ptr, g_ptr, m_ptr, v_ptr, int(n), float(lr), float(beta1), float(beta2),
                                             float(eps), float(weight_decay), int(step), _stream()), "adam_step")


def score_workspace(B, U, I, S, K, topk_only=False, d=None):
    """Bytes of score_topk's workspace. topk_only: the call will ask for top-K lists only (no score matrix). With the
    recdim `d` given the library itself decides whether that call takes the chunked form (no [B x I] block) and returns
    the layout it will use -- the full one for a recdim / K / scorer switch outside the chunked form's range -- with room
    for one chunk's bf16 piece planes (the default scorer of recdim 32 / 64, for lists and for score matrices alike)."""
    if d is not None:
        return int(_lib.load().elimrec_score_workspace_for(B, U, I, S, K, int(d), 0 if topk_only else 1))
    if topk_only:
        return int(_lib.load().elimrec_score_workspace_topk(B, U, I, S, K))
    return int(_lib.load().elimrec_score_workspace2(B, U, I, S, K))


def row_sqnorms(Y, d, n_blocks, out):
    y, ldy = _rowmajor(Y, "Y")
    assert out.is_contiguous() and out.shape == (Y.shape[0], n_blocks)
    _lib.check(_lib.load().elimrec_row_sqnorms(y, ldy, Y.shape[0], d, n_blocks, _dev(out, "out"), _stream()), "row_sqnorms")
    return out


FUSION_MODES = {"rubi": 0, "hm": 1, "sum": 2}
PREDICT_TYPES = {"TE": 1, "TIE": 2}   # anything else -> 0 ("normal", models/EliMRec.py:113)


TIE_ORDERS = {"id": 0, "reference": 1}   # among equal scores: lowest item id first / the reference's heap order (evaluate.h:26-33)


def score_topk(Y, U, I, users, d, S, head_mask, fusion_mode, predict_type, workspace, scores=None, K=0,
               topk_idx=None, topk_val=None, train_ptr=None, train_items=None, sqnorm=None, tie_order="id"):
    y, ldy = _rowmajor(Y, "Y")
    B = users.numel()
    sp, lds = (None, 0)
    if scores is not None:
        sp, lds = _rowmajor(scores, "scores")
    _lib.check(_lib.load().elimrec_score_topk_ordered(y, ldy, U, I, _dev(users, "users", torch.int64), B, d, S, int(head_mask),
                                                      FUSION_MODES[fusion_mode], PREDICT_TYPES.get(predict_type, 0),
                                                      _dev(sqnorm, "sqnorm"), _dev(train_ptr, "train_ptr", torch.int64),
                                                      _dev(train_items, "train_items", torch.int32), sp, lds, int(K),
                                                      _dev(topk_idx, "topk_idx", torch.int32), _dev(topk_val, "topk_val"),
                                                      _dev(workspace, "workspace", torch.uint8), workspace.numel(),
                                                      TIE_ORDERS[tie_order], _stream()),
               "score_topk")


def topk_reference_order(scores, K, out_idx, out_val=None):
    """Rows of masked scores on the device -> the reference's top-K lists (std::partial_sort_copy's order, evaluate.h:26-33)."""
    sp, lds = _rowmajor(scores, "scores")
    B, I = scores.shape
    assert out_idx.is_contiguous() and out_idx.shape == (B, K) and (out_val is None or (out_val.is_contiguous() and out_val.shape == (B, K)))
    _lib.check(_lib.load().elimrec_topk_reference_order_device(sp, B, I, lds, int(K), _dev(out_idx, "out_idx", torch.int32),
                                                               _dev(out_val, "out_val"), _stream()), "topk_reference_order_device")
    return out_idx, out_val


def score_range_violations(reset=True):
    """Waves of the scorer launches since the last reset that saw a score outside their launch's range invariant (a host
    synchronisation with the current stream)."""
    n = ctypes.c_int64(0)
    _lib.check(_lib.load().elimrec_score_range_violations(ctypes.byref(n), 1 if reset else 0, _stream()), "score_range_violations")
    return int(n.value)


def score_range_check(topk_val, topk_idx, predict_type, fusion_mode, row_mean=None):
    """The range check every scoring call ends with, over lists the caller holds (counted into score_range_violations)."""
    B, K = topk_val.shape
    assert topk_val.is_contiguous() and topk_idx.is_contiguous() and topk_idx.shape == (B, K)
    _lib.check(_lib.load().elimrec_score_range_check(_dev(topk_val, "topk_val"), _dev(topk_idx, "topk_idx", torch.int32), B, K,
                                                     PREDICT_TYPES.get(predict_type, 0), FUSION_MODES[fusion_mode],
                                                     _dev(row_mean, "row_mean"), _stream()), "score_range_check")


def score_topk_shard(Y, U, I, users, d, S, head_mask, fusion_mode, predict_type, workspace, phase, row_sum, I_total, id_offset,
                     scores=None, K=0, topk_idx=None, topk_val=None, train_ptr=None, train_items=None, sqnorm=None):
    """elimrec_score_topk_shard: Y = [all user rows ; this shard's I item rows]. phase 1 -> row_sum [B] (TIE), phase 2 ->
    scores [B x I] and / or top-K (catalogue ids) given the all-reduced row_sum."""
    y, ldy = _rowmajor(Y, "Y")
    B = users.numel()
    sp, lds = (None, 0)
    if scores is not None:
        sp, lds = _rowmajor(scores, "scores")
    _lib.check(_lib.load().elimrec_score_topk_shard(y, ldy, U, I, _dev(users, "users", torch.int64), B, d, S, int(head_mask),
                                                    FUSION_MODES[fusion_mode], PREDICT_TYPES.get(predict_type, 0),
                                                    _dev(sqnorm, "sqnorm"), _dev(train_ptr, "train_ptr", torch.int64),
                                                    _dev(train_items, "train_items", torch.int32), sp, lds, int(K),
                                                    _dev(topk_idx, "topk_idx", torch.int32), _dev(topk_val, "topk_val"),
                                                    _dev(workspace, "workspace", torch.uint8), workspace.numel(), int(phase),
                                                    _dev(row_sum, "row_sum"), int(I_total), int(id_offset), _stream()),
               "score_topk_shard")


def topk_merge(cand_val, cand_idx, K, out_idx, out_val=None):
    """[B x n] candidate (value, id) lists -> the K best per row by (score desc, id asc)."""
    B, n = cand_val.shape
    assert cand_val.is_contiguous() and cand_idx.is_contiguous() and cand_idx.shape == (B, n) and out_idx.shape == (B, K)
    _lib.check(_lib.load().elimrec_topk_merge(_dev(cand_val, "cand_val"), _dev(cand_idx, "cand_idx", torch.int32), B, n, int(K),
                                              _dev(out_idx, "out_idx", torch.int32), _dev(out_val, "out_val"), _stream()), "topk_merge")


def rank_metrics(topk_idx, truth_ptr, truth_items, metric_ids, out):
    B, K = topk_idx.shape
    ids = (ctypes.c_int * len(metric_ids))(*metric_ids)
    assert topk_idx.is_contiguous() and out.is_contiguous()
    _lib.check(_lib.load().elimrec_rank_metrics(_dev(topk_idx, "topk_idx", torch.int32), B, K,
                                                _dev(truth_ptr, "truth_ptr", torch.int64),
                                                _dev(truth_items, "truth_items", torch.int32), ids, len(metric_ids),
                                                _dev(out, "out"), _stream()), "rank_metrics")
    return out


def sample_triplets(user_ids, ptr, items, num_items, n, seed, epoch, users, pos, neg):
    _lib.check(_lib.load().elimrec_sample_triplets(_dev(user_ids, "user_ids", torch.int32), _dev(ptr, "ptr", torch.int64),
                                                   _dev(items, "items", torch.int32), user_ids.numel(), num_items, n,
                                                   int(seed), int(epoch), _dev(users, "users", torch.int64),
                                                   _dev(pos, "pos", torch.int64), _dev(neg, "neg", torch.int64),
                                                   _stream()), "sample_triplets")


def head_pack_floats(dims):
    arr = (ctypes.c_int * max(len(dims), 1))(*dims)
    return int(_lib.load().elimrec_head_pack_floats(len(dims), arr))


def head_pack_bwd_offset(dims):
    arr = (ctypes.c_int * max(len(dims), 1))(*dims)
    return int(_lib.load().elimrec_head_pack_bwd_offset(len(dims), arr))


def head_fwd_fused_rows(rows, act, seg_info, c, S, Wm, bm, Wf_user, bf_user, Wf_item, bf_item, Ws, bs, pack, OutAct, YAct, d):
    """elimrec_head_fwd_fused_rows: phase 4 of head_fwd_fused with the rows launch folded in. rows: dict(plan, ns, w, L, U,
    layers (L + 1 flat tensors, the last may be None), long_tab, narrow). Returns (ok, the host struct the call read -- kept by
    the caller for as long as a recorded program refers to it)."""
    n = len(S)
    R = act.numel()
    hr = _lib.HeadRows()
    hr.A = ctypes.pointer(rows["plan"].desc)
    hr.ns, hr.w, hr.L, hr.U = int(rows["ns"]), int(rows["w"]), int(rows["L"]), int(rows["U"])
    for k, t in enumerate(rows["layers"]):
        hr.layers[k] = None if t is None else _dev(t, "layer")
    hr.d_long = _dev(rows["long_tab"], "long_tab")
    nar = rows["narrow"]
    assert nar.stride(1) == 1
    hr.d_narrow_out, hr.ld_narrow_out = _dev(nar, "narrow"), nar.stride(0)
    ptr = lambda ts: (ctypes.c_void_p * max(n, 1))(*[_dev(t, "table") for t in ts])
    ldS = (ctypes.c_int64 * max(n, 1))(*[t.stride(0) for t in S])
    D = (ctypes.c_int * max(n, 1))(*[t.shape[1] for t in S])
    rc = _lib.load().elimrec_head_fwd_fused_rows(
        ctypes.byref(hr), _dev(act, "act", torch.int32), _dev(seg_info, "seg_info", torch.int32), R, _dev(c, "c"), n, ptr(S), ldS, D,
        ptr(Wm), ptr(bm), _dev(Wf_user, "Wf_user"), _dev(bf_user, "bf_user"), _dev(Wf_item, "Wf_item"), _dev(bf_item, "bf_item"),
        ptr(Ws), ptr(bs), _dev(pack, "pack"), pack.numel(), _dev(OutAct, "OutAct"), OutAct.stride(0), _dev(YAct, "YAct"),
        YAct.stride(0), int(d), _stream())
    if rc == 10002:           # ELIMREC_E_UNSUPPORTED
        return False
    _lib.check(rc, "head_fwd_fused_rows")
    return True


def head_fwd_fused(act, seg_info, out0, narrow, c, S, Wm, bm, Wf_user, bf_user, Wf_item, bf_item, Ws, bs, pack, OutAct, YAct, d, phase=0,
                   peers=None):
    """elimrec_head_fwd_fused: S / Wm / bm / Ws / bs are lists over the feature tables. phase 0: pack the weights and
    run the head; 1: pack only; 2: head only (pack holds the packed weights); 3 / 4: the head in two launches (the feature
    blocks without the shared part -- no out0 / narrow needed --, then the rest). Returns False when the shape is outside the
    fused kernel's range (the caller keeps the batched GEMMs). peers: the forward exchange's received buffer [W, R, 2 * dl]
    (out0 | narrow pieces of every peer) read in place of out0 / narrow (elimrec_head_fwd_fused_peers)."""
    n = len(S)
    R = act.numel()
    ptr = lambda ts: (ctypes.c_void_p * max(n, 1))(*[_dev(t, "table") for t in ts])
    ldS = (ctypes.c_int64 * max(n, 1))(*[t.stride(0) for t in S])
    D = (ctypes.c_int * max(n, 1))(*[t.shape[1] for t in S])
    for w in list(Wm) + list(Ws) + [Wf_user, Wf_item]:
        assert w.is_contiguous()
    if peers is not None:
        assert peers.dim() == 3 and peers.is_contiguous() and peers.shape[1] == R and peers.shape[2] % 2 == 0
        rc = _lib.load().elimrec_head_fwd_fused_peers(
            _dev(act, "act", torch.int32), _dev(seg_info, "seg_info", torch.int32), R, _dev(peers, "peers"), peers.shape[0],
            peers.shape[2] // 2, _dev(c, "c"), n, ptr(S), ldS, D, ptr(Wm), ptr(bm), _dev(Wf_user, "Wf_user"),
            _dev(bf_user, "bf_user"), _dev(Wf_item, "Wf_item"), _dev(bf_item, "bf_item"), ptr(Ws), ptr(bs), _dev(pack, "pack"),
            pack.numel(), _dev(OutAct, "OutAct"), OutAct.stride(0), _dev(YAct, "YAct"), YAct.stride(0), int(d), int(phase), _stream())
        if rc == 10002:
            return False
        _lib.check(rc, "head_fwd_fused_peers")
        return True
    rc = _lib.load().elimrec_head_fwd_fused(
        _dev(act, "act", torch.int32), _dev(seg_info, "seg_info", torch.int32), R, _dev(out0, "out0"), out0.stride(0),
        _dev(narrow, "narrow"), narrow.stride(0), _dev(c, "c"), n, ptr(S), ldS, D, ptr(Wm), ptr(bm), _dev(Wf_user, "Wf_user"),
        _dev(bf_user, "bf_user"), _dev(Wf_item, "Wf_item"), _dev(bf_item, "bf_item"), ptr(Ws), ptr(bs), _dev(pack, "pack"),
        pack.numel(), _dev(OutAct, "OutAct"), OutAct.stride(0), _dev(YAct, "YAct"), YAct.stride(0), int(d), int(phase), _stream())
    if rc == 10002:           # ELIMREC_E_UNSUPPORTED
        return False
    _lib.check(rc, "head_fwd_fused")
    return True


def head_fwd_fused_src16(fshard, S_out, c_out, act, seg_info, out0, narrow, Wm, bm, Wf_user, bf_user, Wf_item, bf_item, Ws, bs, pack,
                         OutAct, YAct, d, phase=0):
    """elimrec_head_fwd_fused_src16: head_fwd_fused with the feature constants read from `fshard`'s 16-bit rows (lookup.FeatureShard
    holding EVERY row: one rank); S_out [R x sum_d] / c_out [R] (or None) receive the widened rows of the active nodes."""
    n = len(fshard.dims)
    R = act.numel()
    src = _lib.HeadSrc16()
    src.d_table, src.row_elems, src.dtype = fshard.table.data_ptr(), fshard.row_elems, fshard.code
    if S_out is not None:
        assert S_out.stride(1) == 1 and S_out.shape[0] >= R and S_out.shape[1] == fshard.sum_d and c_out.numel() >= R
        src.d_S_out, src.ld_S_out, src.d_c_out = _dev(S_out, "S_out"), S_out.stride(0), _dev(c_out, "c_out")
    ptr = lambda ts: (ctypes.c_void_p * max(n, 1))(*[_dev(t, "table") for t in ts])
    D = (ctypes.c_int * max(n, 1))(*fshard.dims)
    rc = _lib.load().elimrec_head_fwd_fused_src16(
        ctypes.byref(src), _dev(act, "act", torch.int32), _dev(seg_info, "seg_info", torch.int32), R, _dev(out0, "out0"), out0.stride(0),
        _dev(narrow, "narrow"), narrow.stride(0), n, D, ptr(Wm), ptr(bm), _dev(Wf_user, "Wf_user"), _dev(bf_user, "bf_user"),
        _dev(Wf_item, "Wf_item"), _dev(bf_item, "bf_item"), ptr(Ws), ptr(bs), _dev(pack, "pack"), pack.numel(), _dev(OutAct, "OutAct"),
        OutAct.stride(0), _dev(YAct, "YAct"), YAct.stride(0), int(d), int(phase), _stream())
    if rc == 10002:           # ELIMREC_E_UNSUPPORTED
        return False
    _lib.check(rc, "head_fwd_fused_src16")
    return True


def peer_cols_to_rows(recv, out0, out1):
    """recv [W x R x 2*dl] (per peer: layer mean | shared part of my rows in its columns) -> out0 / out1 [R x W*dl] row views."""
    W, R, two_dl = recv.shape
    dl = two_dl // 2
    assert recv.is_contiguous() and out0.stride(1) == 1 and out1.stride(1) == 1 and out0.shape == (R, W * dl) and out1.shape == (R, W * dl)
    _lib.check(_lib.load().elimrec_peer_cols_to_rows(_dev(recv, "recv"), W, R, dl, _dev(out0, "out0"), out0.stride(0), _dev(out1, "out1"),
                                                    out1.stride(0), _stream()), "peer_cols_to_rows")
