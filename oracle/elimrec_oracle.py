"""CPU oracle for the EliMRec hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module. The product path (elimrec_amd/) never does: it fails loudly without the HIP
library instead of falling back to anything in here.

What it is: a plain torch-CPU fp32 restatement of the reference's per-batch loop, each
function citing the reference lines it follows (paths relative to /root/reference).
Pinned: tests/test_oracle_golden.py checks every function below against fixtures captured
by importing the reference itself (tests/golden/make_golden.py; tests/golden/*.npz).

Differences from the reference that are deliberate and do not change results:
  * the model is a bag of tensors keyed by the reference's state_dict names, not an
    nn.Module; autograd still provides the backward pass (as in the reference);
  * the adjacency is built once with scipy exactly as the reference does and handed to
    torch.sparse.mm as a COO tensor (reference: models/EliMRec.py:80-84,244).
"""
import math

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn.functional as F

EPS = 1e-12  # models/EliMRec.py:13


# --------------------------------------------------------------------------- adjacency
def build_adj(train_u, train_i, num_users, num_items, adj_type="pre"):
    """models/EliMRec.py:309-354 (create_adj_mat). Returns scipy CSR float32 [N,N]."""
    user_np = np.asarray(train_u, dtype=np.int32)
    item_np = np.asarray(train_i, dtype=np.int32)
    ratings = np.ones_like(user_np, dtype=np.float32)
    n = num_users + num_items
    tmp = sp.csr_matrix((ratings, (user_np, item_np + num_users)), shape=(n, n))
    adj = tmp + tmp.T

    def normalized_adj_single(a):                     # :319-327
        rowsum = np.array(a.sum(1))
        with np.errstate(divide="ignore"):
            d_inv = np.power(rowsum, -1).flatten()
        d_inv[np.isinf(d_inv)] = 0.0
        return sp.diags(d_inv).dot(a).tocoo()

    if adj_type == "plain":                           # :329-331
        m = adj
    elif adj_type == "norm":                          # :332-335
        m = normalized_adj_single(adj + sp.eye(adj.shape[0]))
    elif adj_type == "gcmc":                          # :336-338
        m = normalized_adj_single(adj)
    elif adj_type == "pre":                           # :339-348
        rowsum = np.array(adj.sum(1))
        with np.errstate(divide="ignore"):
            d_inv = np.power(rowsum, -0.5).flatten()
        d_inv[np.isinf(d_inv)] = 0.0
        d = sp.diags(d_inv)
        m = d.dot(adj).dot(d)
    else:                                             # :349-352
        mean_adj = normalized_adj_single(adj)
        m = mean_adj + sp.eye(mean_adj.shape[0])
    return sp.csr_matrix(m).astype(np.float32)


def adj_to_torch(adj_csr):
    """models/EliMRec.py:80-84: COO, int64 indices, fp32 values."""
    coo = adj_csr.tocoo()
    idx = torch.from_numpy(np.vstack([coo.row, coo.col]).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(coo.data.astype(np.float32)), coo.shape).coalesce()


# --------------------------------------------------------------------------- model
class OracleEliMRec:
    """State + forward of models/EliMRec.py. `params` uses the reference's state_dict keys."""

    def __init__(self, num_users, num_items, recdim, layer_num, adj, feats, params, alpha,
                 dataset_name="movielens", modality="vat", mm_fusion_mode="concat",
                 fusion_mode="rubi", predict_type="TIE", mods=None, dtype=torch.float32, words=None):
        """dtype: torch.float32 is the reference's arithmetic (and what every fixture pins). torch.float64 evaluates the
        SAME formulas on the same fp32 inputs in double precision: the tests use it to tell a row where the fp32 reference
        itself is off by more than the tolerance (cancellation) from a row where the HIP path is."""
        self.U, self.I, self.d, self.L = int(num_users), int(num_items), int(recdim), int(layer_num)
        self.adj = adj if isinstance(adj, torch.Tensor) else adj_to_torch(adj)
        if dtype != torch.float32:
            self.adj = self.adj.to(dtype)
        self.kwai = dataset_name == "kwai"              # EliMRec.py:133,148,158,234,254,261
        # `mods`: NOT in the reference (it hard-codes V for kwai, V,A,T otherwise). A generalisation used only
        # for BASELINE.json's "V+T" Kwai-shape case; parity for it is unpinned (no reference run exists).
        self.mods = list(mods) if mods is not None else (["v"] if self.kwai else ["v", "a", "t"])
        self.feats = {k: torch.as_tensor(v, dtype=torch.float32).to(dtype) for k, v in feats.items()}
        self.params = {k: torch.as_tensor(np.array(v), dtype=torch.float32).to(dtype).clone().requires_grad_(True)
                       for k, v in params.items()}
        # models/EliMRec.py:371-378 (data set "tiktok"): t_feat = scatter-mean of the word embeddings of every item's words (`words`
        # = dataset.words_tensor, [2 x n]: item id, word id), built ONCE from the initial word_embedding.weight, not normalised --
        # and left attached to it: with main.py:100's backward(retain_graph=True) the parameter receives the gradient that
        # arrives at t_feat every step (and coupled weight decay), although its new values never reach a forward pass again.
        self.retain_graph = False
        if words is not None:
            w = self.params["word_embedding.weight"]
            idx0, idx1 = (torch.as_tensor(np.asarray(x), dtype=torch.int64) for x in words)
            n = int(idx0.max()) + 1
            tot = torch.zeros(n, w.shape[1], dtype=w.dtype).index_add(0, idx0, w[idx1])
            cnt = torch.zeros(n, dtype=w.dtype).index_add(0, idx0, torch.ones(idx0.numel(), dtype=w.dtype)).clamp(min=1)
            self.feats["t"] = tot / cnt[:, None]
            self.retain_graph = True
        self.alpha = float(alpha)
        self.modality = "v" if self.kwai else modality  # EliMRec.py:133-134
        self.mm_fusion_mode = mm_fusion_mode
        self.fusion_mode = fusion_mode
        self.predict_type = predict_type
        self.all_users = self.all_items = None
        self.all_s_embs = None

    @staticmethod
    def normalize_features(raw):
        """models/EliMRec.py:366-381: F.normalize(feat.float(), dim=1)."""
        return F.normalize(torch.as_tensor(raw).float(), dim=1)

    def _linear(self, name, x):
        return F.linear(x, self.params[name + ".weight"], self.params[name + ".bias"])

    def _compute_graph(self, u_emb, i_emb):
        """models/EliMRec.py:238-248."""
        all_emb = torch.cat([u_emb, i_emb])
        embs = [all_emb]
        for _ in range(self.L):
            all_emb = torch.sparse.mm(self.adj, all_emb)
            embs.append(all_emb)
        return torch.mean(torch.stack(embs, dim=1), dim=1)

    def _mm_fusion(self, reps):
        """models/EliMRec.py:221-226."""
        if self.mm_fusion_mode == "concat":
            return torch.cat(reps, dim=1)
        return torch.mean(torch.stack(reps), dim=0)

    def compute(self):
        """models/EliMRec.py:228-272."""
        p = self.params
        users_emb = p["embedding_user.weight"]
        items_emb = p["embedding_item.weight"]
        mods = self.mods
        dense = {m: self._linear("%s_dense" % m, self.feats[m]) for m in mods}   # :233-236
        self.m_emb = {"i": self._compute_graph(users_emb, items_emb)}           # :250
        for m in mods:                                                           # :252-256
            self.m_emb[m] = self._compute_graph(users_emb, dense[m])
        split = lambda x: torch.split(x, [self.U, self.I])
        us = [split(self.m_emb[k])[0] for k in ["i"] + mods]
        its = [split(self.m_emb[k])[1] for k in ["i"] + mods]
        user = self._linear("embedding_user_after_GCN", self._mm_fusion(us))    # :262-270
        item = self._linear("embedding_item_after_GCN", self._mm_fusion(its))
        return user, item

    def gcn_cf(self):
        """models/EliMRec.py:144-153."""
        out = {}
        for m in self.mods:
            e = self._linear("s_dense_%s" % m, self.m_emb[m])
            out["pre_fusion_user_" + m], out["pre_fusion_item_" + m] = torch.split(e, [self.U, self.I])
        return out

    @staticmethod
    def original_bpr_loss(u, p, n):
        """models/EliMRec.py:291-297."""
        u = F.normalize(u, dim=1)
        p = F.normalize(p, dim=1)
        n = F.normalize(n, dim=1)
        return torch.mean(F.softplus(torch.sum(u * n, dim=1) - torch.sum(u * p, dim=1)))

    def bpr_loss(self, users, pos, neg):
        """models/EliMRec.py:115-142 (+ getEmbedding :274-289)."""
        users = torch.as_tensor(users).long()
        pos = torch.as_tensor(pos).long()
        neg = torch.as_tensor(neg).long()
        self.all_users, self.all_items = self.compute()
        self.all_s_embs = self.gcn_cf()
        fusion = self.original_bpr_loss(self.all_users[users], self.all_items[pos], self.all_items[neg])
        if self.predict_type == "normal":                                       # :125-126
            return fusion
        p_loss = 0
        for m in self.modality:                                                  # :136-140
            if m not in self.mods:
                continue
            s = self.all_s_embs
            p_loss = p_loss + self.original_bpr_loss(s["pre_fusion_user_" + m][users],
                                                     s["pre_fusion_item_" + m][pos],
                                                     s["pre_fusion_item_" + m][neg])
        return fusion + self.alpha * p_loss                                      # :142

    # ---- evaluation-time scoring -------------------------------------------------
    def general_cm_fusion(self, fusion_logits, users):
        """models/EliMRec.py:155-212 with items=None, normalize=True."""
        s = self.all_s_embs
        mods = self.mods
        z = {}
        for m in mods:
            su = F.normalize(s["pre_fusion_user_" + m][users], dim=1)
            si = F.normalize(s["pre_fusion_item_" + m], dim=1)
            z[m] = torch.matmul(su, si.t())
        if self.fusion_mode == "rubi":                                           # :171-188
            out = fusion_logits
            for m in mods:
                if m in self.modality:
                    out = out * torch.sigmoid(z[m])
            return out
        if self.fusion_mode == "hm":                                             # :190-199
            out = torch.sigmoid(fusion_logits)
            for m in mods:
                out = out * torch.sigmoid(z[m])
            return torch.log(out + EPS) - torch.log1p(out)
        if self.fusion_mode == "sum":                                            # :201-210
            out = fusion_logits
            for m in mods:
                out = out + z[m]
            return torch.log(torch.sigmoid(out) + EPS)
        raise ValueError(self.fusion_mode)

    def predict(self, user_ids):
        """models/EliMRec.py:96-113. Uses the tables cached by the last bpr_loss() call."""
        with torch.no_grad():
            users = torch.as_tensor(np.asarray(user_ids)).long()
            ui = torch.sigmoid(torch.matmul(self.all_users[users], self.all_items.t()))
            if self.predict_type == "TE":
                return torch.sigmoid(self.general_cm_fusion(ui, users))
            if self.predict_type == "TIE":
                te = self.general_cm_fusion(ui, users)
                nde = self.general_cm_fusion(torch.mean(ui, -1, True), users)
                return torch.sigmoid(te - nde)
            return torch.sigmoid(ui)

    def set_cache(self, all_users, all_items, s_embs):
        self.all_users = torch.as_tensor(all_users)
        self.all_items = torch.as_tensor(all_items)
        self.all_s_embs = {k: torch.as_tensor(v) for k, v in s_embs.items()}

    def grads(self):
        return {k: v.grad for k, v in self.params.items() if v.grad is not None}

    def zero_grad(self):
        for v in self.params.values():
            v.grad = None


# --------------------------------------------------------------------------- optimiser
class OracleAdam:
    """torch.optim.Adam as main.py:49,101 uses it: betas (0.9,0.999), eps 1e-8, coupled L2
    (`grad += weight_decay * param`), dense over every parameter that has a gradient.
    Written out explicitly (single-tensor, non-capturable path of torch/optim/adam.py)."""

    def __init__(self, params, lr=1e-3, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.params, self.lr, self.wd, self.b1, self.b2, self.eps = params, lr, weight_decay, betas[0], betas[1], eps
        self.state = {}

    @torch.no_grad()
    def step(self):
        for k, p in self.params.items():
            if p.grad is None:
                continue
            st = self.state.setdefault(k, {"t": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)})
            st["t"] += 1
            t = st["t"]
            g = p.grad
            if self.wd != 0:
                g = g.add(p, alpha=self.wd)
            st["m"].lerp_(g, 1 - self.b1)
            st["v"].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            bc1 = 1 - self.b1 ** t
            bc2 = 1 - self.b2 ** t
            step_size = self.lr / bc1
            denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(st["m"], denom, value=-step_size)


def train_step(model, opt, users, pos, neg):
    """main.py:98-102 loop body."""
    loss = model.bpr_loss(users, pos, neg)
    model.zero_grad()
    loss.backward(retain_graph=bool(getattr(model, "retain_graph", False)))      # main.py:100
    opt.step()
    return float(loss.item())
