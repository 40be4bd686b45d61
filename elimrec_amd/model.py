"""EliMRec on MI355X: the reference's plugin class (models/EliMRec.py:31-407) with the same
constructor, parameter names, state_dict keys and methods, whose per-batch work runs entirely
in libelimrec_hip.so.

Data layout in HBM (fp32, row-major; N = U + I nodes, users first):
  X0/Out [N x C], C = M*d   every propagated table side by side: column block 0 = id table,
                            1.. = projected V/A/T features. One 1-KiB row (Tiktok shape) serves
                            all M LightGCN graphs, so the CSR structure is read once per hop.
  Y      [N x Cy], Cy=(1+S)d block 0 = fused embedding (all_users / all_items), block 1+h =
                            single-modal head h (pre_fusion_{user,item}_{v,a,t}). The BPR head
                            gathers one contiguous row per index; predict() reads only Y.
What the reference does with 12+12 torch.sparse.mm calls, 8 addmm and autograd per batch is here:
  default (bipartite adjacency, or --propagation=folded on one with a diagonal; layer_num >= 2): the constant feature tables
      folded into GEMM operands, only the d-column id table through the graph, everything after the graph at the batch's
      active rows -- the column-shard engine (shard.py: slab-major tables, wave-tile hops, fused head, Adam as the last
      adjoint hop's epilogue). bpr_loss -> backward -> optimizer.step completes a request the engine runs as ONE enqueue
      (plugin.py). The full cached tables predict() reads are materialised on first use (_ensure_tables).
  the unfolded forms (any adjacency: --propagation=full | bipartite, --head_rows=all, layer_num < 2), row-major, every row
      every step: assemble_x0 + 3 linear_fwd -> propagate -> 5 linear_fwd -> bpr_head, and back: segment plan/apply ->
      head_bwd_input / linear_bwd_w -> propagate (A^T) -> embed_grad / linear_bwd_w. Also what compute() and the generic
      BasicModel losses differentiate through, and bench.py's "reference-equivalent work" line.
Propagation is linear, so nothing of the forward pass is kept for backward except Out (or its active rows) and Y.
"""
import os

import numpy as np
import scipy.sparse as sp
import torch
from torch import nn
from torch.nn import functional as F

from . import _lib, ops
from .basic_model import BasicModel
from .logger import Logger
from .plugin import EmbeddingParameter, LazyGradParameter, StepController


def create_adj_mat(train_users, train_items, num_users, num_items, adj_type):
    """models/EliMRec.py:309-354. Returns a scipy CSR [N, N] in fp32 (values rounded as scipy does)."""
    u = np.asarray(train_users, dtype=np.int32)
    i = np.asarray(train_items, dtype=np.int32)
    n = num_users + num_items
    half = sp.csr_matrix((np.ones_like(u, dtype=np.float32), (u, i + num_users)), shape=(n, n))
    adj = half + half.T

    def row_normalised(a):
        deg = np.asarray(a.sum(1)).flatten()
        with np.errstate(divide="ignore"):
            inv = np.power(deg, -1.0)
        inv[np.isinf(inv)] = 0.0
        return sp.diags(inv).dot(a)

    if adj_type == "plain":
        out = adj
    elif adj_type == "norm":
        out = row_normalised(adj + sp.eye(n))
    elif adj_type == "gcmc":
        out = row_normalised(adj)
    elif adj_type == "pre":
        deg = np.asarray(adj.sum(1))
        with np.errstate(divide="ignore"):
            inv = np.power(deg, -0.5).flatten()
        inv[np.isinf(inv)] = 0.0
        dm = sp.diags(inv)
        out = dm.dot(adj).dot(dm)
    else:
        out = row_normalised(adj) + sp.eye(n)
    out = sp.csr_matrix(out).astype(np.float32)
    out.sort_indices()
    return out


class _BprLossFn(torch.autograd.Function):
    """Glue between torch's autograd/optimizer API (main.py:98-101) and the HIP path: forward enqueues the forward
    kernels and returns the 0-dim loss; backward enqueues the backward kernels and leaves every parameter's gradient in
    `.grad` (nothing for parameters the loss does not reach: the set the reference leaves without .grad -- with ONE
    stated exception, `word_embedding.weight` on the tiktok data set: the reference builds t_feat once from it and keeps
    that graph alive with retain_graph=True (main.py:99), so the dead parameter keeps receiving a gradient and an Adam /
    weight-decay update every step; here t_feat is a constant and word_embedding a checkpoint key that stays at its
    initial value -- losses and scores are unaffected, the saved tensor differs from a reference run's).
    The gradients live in one flat buffer laid out like the parameters; backward assigns its views to `.grad` itself
    instead of returning them to autograd, whose AccumulateGrad would copy each of the 18 tensors (the views are not
    "stealable") -- 18 clones per step, and an optimizer that no longer sees adjacent gradients. A `.grad` that already
    holds something else is accumulated into, as autograd would; a `.grad` that already IS the flat view is overwritten
    (two backward passes without zero_grad do not accumulate -- the reference's loop zeroes every step, main.py:98).
    The forward's state lives in the model's workspace, not in the autograd context: backward must follow ITS forward.
    Each forward takes a generation number; backward raises if another forward (or compute()) ran in between."""

    @staticmethod
    def forward(ctx, model, users, pos, neg, *params):
        ctx.model = model
        ctx.params = params
        ctx.names = model._param_names
        loss = model._forward_hip(users, pos, neg, need_grad=True)
        ctx.gen = model._fwd_gen
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        model = ctx.model
        if ctx.gen != model._fwd_gen:
            raise RuntimeError("backward of a stale loss: another forward ran on this model since (the forward's state "
                               "lives in the model's workspace; call backward before the next bpr_loss / compute)")
        # upstream gradient into a fixed buffer: the backward regions are recorded against stable addresses
        gscale = model._ws.setdefault("gscale", torch.ones(1, dtype=torch.float32, device=grad_out.device))
        gscale.copy_(grad_out.reshape(1))
        grads = model._backward_hip(gscale)
        for name, p in zip(ctx.names, ctx.params):
            g = grads.get(name)
            if g is None or not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = g
            elif p.grad.data_ptr() != g.data_ptr():
                p.grad.add_(g)
        return (None,) * (4 + len(ctx.params))


class _TablesFn(torch.autograd.Function):
    """compute() (models/EliMRec.py:228-272) as a differentiable op for losses written against the tables themselves
    (BasicModel.bpr_loss / infonce / fast_loss on top of getEmbedding, models/BasicModel.py:59-113): forward runs the
    HIP table build and returns copies of all_users / all_items; backward takes DENSE table gradients, treats every node
    as an active row of the fused head block and runs the same HIP backward as the training step. Parameter gradients
    land in `.grad` (the flat gradient views) as with _BprLossFn."""

    @staticmethod
    def forward(ctx, model, *params):
        ctx.model, ctx.params, ctx.names = model, params, model._param_names
        ws = model._workspace(model._ws_key[1] if model._ws_key else 1, parity=model._ws_key[3] if model._ws_key else 0)
        model._slab_fwd = False
        model._fwd_gen = getattr(model, "_fwd_gen", 0) + 1
        ctx.gen = model._fwd_gen
        model._compute_tables(ws)
        U, d = model.num_users, model.latent_dim
        return ws["Y"][:U, :d].clone(), ws["Y"][U:, :d].clone()

    @staticmethod
    def backward(ctx, g_users, g_items):
        model = ctx.model
        if ctx.gen != model._fwd_gen:
            raise RuntimeError("backward of stale tables: another forward ran on this model since compute()")
        grads = model._backward_tables(g_users, g_items)
        for name, p in zip(ctx.names, ctx.params):
            g = grads.get(name)
            if g is None or not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = g
            elif p.grad.data_ptr() != g.data_ptr():
                p.grad.add_(g)
        return (None,) * (1 + len(ctx.params))


class EliMRec(BasicModel):
    def __init__(self, config, dataset):
        super(EliMRec, self).__init__(dataset, config)
        self.__init_weight()

    # ------------------------------------------------------------------ construction
    def __init_weight(self):
        cfg = self.config
        self.num_users = self.dataset.num_users
        self.num_items = self.dataset.num_items
        self.latent_dim = cfg["recdim"]
        self.n_layers = cfg["layer_num"]
        self.temp = cfg["temp"]
        self.logits = cfg["logits"]
        opt = lambda key, default: cfg[key] if key in cfg else default
        self.predict_type = opt("predict_type", "TIE")
        Logger.info("predict type: " + self.predict_type)
        self.mm_fusion_mode = opt("mm_fusion_mode", "concat")
        Logger.info("mm fusion mode: " + self.mm_fusion_mode)
        self.is_u_s = opt("is_u_s", False)
        Logger.info("cf in single modal preference: " + str(self.is_u_s))
        self.fusion_mode = opt("s_fusion_mode", "rubi")
        Logger.info("score fusion mode: " + self.fusion_mode)
        Logger.info("alpha: " + str(cfg["alpha"]))          # KeyError if missing, as in the reference (:66)
        self.modified_ui_loss = opt("modified_ui_loss", True)
        Logger.info("modified_ui_loss: " + str(self.modified_ui_loss))
        self.modality = opt("modality", "vat")
        Logger.info("Modality Ablation: " + str(self.modality))
        # --lean_tables=1 (CLI-only; BASELINE.json configs[4]): nothing of size N lives on the device outside the
        # column-sharded engine -- the embedding PARAMETERS and the raw feature tables stay on the host (the engine uploads
        # its column slice / its item block), no device CSR copies of the graph, no [N x C] tables in the workspace. Training
        # and evaluation run through ColumnShardTrainer / the item-sharded evaluator only.
        self._lean = str(opt("lean_tables", "0")) in ("1", "True", "true")
        if self.latent_dim % 4 != 0:
            raise ValueError("recdim must be a multiple of 4 for the HIP kernels (got %d)" % self.latent_dim)
        if self.mm_fusion_mode not in ("concat", "mean"):
            raise ValueError("mm_fusion_mode must be 'concat' or 'mean'")
        self.dataset_name = cfg["data.input.dataset"]
        self.is_kwai = self.dataset_name == "kwai"
        # tables after the id table. The reference keys this on the data set name only (kwai -> V, else V,A,T:
        # EliMRec.py:148,234,254,261); `--feature_modalities=vt` (CLI-only, not in the reference) selects any
        # subset whose features the data set provides, e.g. BASELINE.json's Kwai-shape "V+T" configuration.
        self._mods = ["v"] if self.is_kwai else ["v", "a", "t"]
        if "feature_modalities" in cfg:
            self._mods = [m for m in "vat" if m in str(cfg["feature_modalities"])]
            if not self._mods or any(not hasattr(self.dataset, m + "_feat") for m in self._mods):
                raise ValueError("feature_modalities=%r needs the matching <m>_feat tensors on the data set"
                                 % cfg["feature_modalities"])
        self.M = 1 + len(self._mods)
        self.C = self.M * self.latent_dim
        self.S = len(self._mods)                                       # single-modal heads
        self.Cy = (1 + self.S) * self.latent_dim

        self._create_u_embeding_i()
        self._cache = None                     # (all_users, all_items, all_s_embs) views into Y
        self._tables_dirty = False

        tu, ti = self.dataset.get_train_interactions()
        if str(opt("adj_build", "host")) == "device":     # CLI-only: degree count, normalisation and sort on the GPU (adjacency.py)
            from .adjacency import build_adj_device
            rp, cl, vl = build_adj_device(tu, ti, self.num_users, self.num_items, cfg["adj_type"], torch.device("cuda", torch.cuda.current_device()))
            n_nodes = self.num_users + self.num_items
            adj = sp.csr_matrix((vl.cpu().numpy(), cl.cpu().numpy(), rp.cpu().numpy()), shape=(n_nodes, n_nodes))
        else:
            adj = create_adj_mat(tu, ti, self.num_users, self.num_items, cfg["adj_type"])
        self._adj_host = adj                      # the engine's plan is built from this (host) matrix ...
        self._plan_build = str(opt("plan_build", "host"))      # ... or, --plan_build=device (CLI-only), from the device CSR by csrc/plan.hip
        if self._plan_build not in ("host", "device"):
            raise ValueError("plan_build must be host or device")
        if not self._lean:
            self._register_csr("adj", adj)
        adj_t = adj.T.tocsr()
        adj_t.sort_indices()
        self._adj_symmetric = (abs(adj - adj_t)).nnz == 0
        if not self._adj_symmetric and not self._lean:
            self._register_csr("adjT", adj_t)
        # Bipartite fast path (csrc/spmm.hip): no user-user / item-item entries (true for 'pre', 'plain', 'gcmc')
        U = self.num_users
        mode = opt("propagation", "auto")        # CLI-only: auto | folded | bipartite | full
        if mode not in ("auto", "folded", "bipartite", "full"):
            raise ValueError("propagation must be auto, folded, bipartite or full")
        self._bipartite = (self.n_layers >= 1 and adj[:U, :U].nnz == 0 and adj[U:, U:].nnz == 0 and mode != "full")
        # "folded": the propagated FEATURE tables are linear in the projection weights with constant factors,
        # mean_k A^k [0 ; F_m W_m^T + 1 b_m^T] = (mean_k A^k [0 ; F_m]) W_m^T + (mean_k A^k [0 ; 1]) b_m^T,
        # so the constant matrices S_m = mean_k A^k [0 ; F_m] ([N x D_m]) and c = mean_k A^k [0 ; 1] are
        # propagated ONCE at start-up; per step only the id table and the shared user part go through the
        # graph (d columns instead of M*d) and the feature blocks of Out come from one GEMM each.
        self._folded = self._bipartite and mode in ("auto", "folded")
        # an adjacency WITH a diagonal (adj_type norm / mean + I) and --propagation=folded asked for explicitly: the folded
        # algebra holds for any A-hat, only the one-table parity trick does not -- the column-sharded engine then carries the
        # E_u-borne and the E_i-borne part side by side ("wide" tables, csrc/wide.hip, shard.py). The model's own training path
        # (bpr_loss through autograd) has no such form: train with ColumnShardTrainer (main.py does), evaluate as ever.
        no_cross = adj[:U, :U].nnz == 0 and adj[U:, U:].nnz == 0
        self._wide = (not no_cross) and mode == "folded" and self.n_layers >= 2
        if self._wide:
            self._folded = True
        # "batch" head rows (CLI-only: --head_rows=batch|all): the projections AFTER the graph (feature blocks of
        # Out, embedding_*_after_GCN, s_dense_*) are row-wise, and the loss reads them at the batch's 3B rows only,
        # so a training step evaluates them there; the full cached tables predict() reads (:98-99) are filled in on
        # first use from the graph tables of that same forward and a copy of the (pre-update) projection weights.
        self._lazy = self._folded and self.n_layers >= 2 and str(opt("head_rows", "batch")) != "all"
        if self._lean and not self._lazy:
            raise ValueError("--lean_tables=1 needs the folded propagation with batch head rows (bipartite adjacency, layer_num >= 2)")
        if self._bipartite and not self._lean:
            P, Q = adj[:U, U:].tocsr(), adj[U:, :U].tocsr()
            self._register_csr("bipP", P)
            self._register_csr("bipQ", Q)
            if not self._adj_symmetric:
                self._register_csr("bipPT", P.T.tocsr())
                self._register_csr("bipQT", Q.T.tocsr())

        # same construction order as the reference (:88-93) so a seeded run draws the same init
        d = self.latent_dim
        self.s_dense_v = nn.Linear(d, d)
        self.s_dense_a = nn.Linear(d, d)
        self.s_dense_t = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.s_dense_v.weight)
        nn.init.xavier_uniform_(self.s_dense_a.weight)
        nn.init.xavier_uniform_(self.s_dense_t.weight)
        # the same tensors as parameter classes that know about the deferred training step (plugin.py): `.grad` of every
        # parameter and the VALUE of the two embedding tables become real on first read
        self._plugin = StepController(self)
        for mod in self.modules():
            for pname, prm in list(mod._parameters.items()):
                if prm is not None:
                    cls = EmbeddingParameter if mod in (self.embedding_user, self.embedding_item) else LazyGradParameter
                    mod._parameters[pname] = cls(prm.data, prm.requires_grad)
        self._plugin.adopt(self.parameters())
        self._param_names = [n for n, _ in self.named_parameters()]
        # step regions: recorded C-ABI call lists (see _region)
        self._use_replay = os.environ.get("ELIMREC_REPLAY", "1") != "0"
        self._regions = {}
        self._ws = None
        self._ws_key = None

    def __setattr__(self, name, value):
        # per-step bookkeeping (plan sizes, cache markers, ...) skips nn.Module's parameter / buffer / submodule checks
        if name[0] == "_" and not isinstance(value, (torch.Tensor, torch.nn.Module)) and name not in self.__dict__.get("_buffers", ()):
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    # cached tables of the last forward (models/EliMRec.py:121-122); materialised on first use in "batch" mode
    def _cached(self, k):
        self._plugin.realise_forward()         # a bpr_loss whose step has not run yet IS the last training forward
        if self._cache is None:
            return None
        self._ensure_tables()
        if self.__dict__.get("_eval_shard") is not None and k != 0:
            raise RuntimeError("the cached item tables are sharded over the ranks (--feature_shard=row): all_items / all_s_embs "
                               "exist per item block only; use predict() / evaluate(), which gather what they need")
        if self._cache is True:                 # published by a training step: the views are built on first use
            Y, U, d = self._cache_src, self.num_users, self.latent_dim
            s_embs = {}
            for h, m in enumerate(self._mods):
                blk = Y[:, (h + 1) * d:(h + 2) * d]
                s_embs["pre_fusion_user_" + m] = blk[:U]
                s_embs["pre_fusion_item_" + m] = blk[U:]
            self._cache = (Y[:U, :d], Y[U:, :d], s_embs)
        return self._cache[k]

    all_users = property(lambda self: self._cached(0))
    all_items = property(lambda self: self._cached(1))
    all_s_embs = property(lambda self: self._cached(2))

    def _register_csr(self, name, m):
        m = m.tocsr()
        m.sort_indices()
        if m.nnz >= 2 ** 31 or m.shape[0] >= 2 ** 31 - 1:
            raise ValueError("graph too large for int32 CSR")
        self.register_buffer(name + "_rowptr", torch.from_numpy(m.indptr.astype(np.int32)), persistent=False)
        self.register_buffer(name + "_col", torch.from_numpy(m.indices.astype(np.int32)), persistent=False)
        self.register_buffer(name + "_val", torch.from_numpy(m.data.astype(np.float32)), persistent=False)

    def _scipy_adj(self):
        """The propagation matrix as scipy CSR (the host matrix it was built as)."""
        if self.__dict__.get("_adj_host") is not None:
            return self._adj_host
        n = self.num_users + self.num_items
        return sp.csr_matrix((self.adj_val.cpu().numpy(), self.adj_col.cpu().numpy(), self.adj_rowptr.cpu().numpy()),
                             shape=(n, n))

    def _csr(self, name):
        """Device CSR + its row-split plan (built on first use on the current device)."""
        rowptr = getattr(self, name + "_rowptr")
        cache = self.__dict__.setdefault("_csr_cache", {})
        hit = cache.get(name)
        if hit is None or hit.rowptr.data_ptr() != rowptr.data_ptr():
            hit = ops.Csr(rowptr, getattr(self, name + "_col"), getattr(self, name + "_val"), rowptr.numel() - 1)
            if rowptr.is_cuda:
                hit.build_split(self.C)
            cache[name] = hit
        return hit

    def _create_u_embeding_i(self):
        """models/EliMRec.py:356-407, same RNG draw order."""
        d = self.latent_dim
        self.embedding_user = nn.Embedding(self.num_users, d)
        self.embedding_item = nn.Embedding(self.num_items, d)
        nn.init.xavier_uniform_(self.embedding_user.weight)
        nn.init.xavier_uniform_(self.embedding_item.weight)
        Logger.info("[use Xavier initilizer]")
        ds = self.dataset
        custom = "feature_modalities" in self.config
        if self._lean:
            # host tensors (not buffers: .to(device) leaves them where they are); the engine's distributed fold uploads the
            # rank's item block of each (shard.py)
            if self.dataset_name == "tiktok" and not custom:
                raise ValueError("--lean_tables=1 does not cover the tiktok word-bag text features")
            # --feature_load=block: not even that -- a reader that serves normalised row blocks (dataset.FeatureBlocks); the
            # distributed fold asks it for the rank's item block and for nothing else
            block = str(self.config["feature_load"]) == "block" if "feature_load" in self.config else False
            for m in self._mods:
                object.__setattr__(self, m + "_feat", ds.feature_blocks(m) if block else
                                   F.normalize(getattr(ds, m + "_feat").float(), dim=1).contiguous())
        elif "feature_load" in self.config and str(self.config["feature_load"]) == "block":
            raise ValueError("--feature_load=block needs --lean_tables=1 (and --feature_shard=row): the other forms keep whole feature tables")
        elif "v" in self._mods:
            self.register_buffer("v_feat", F.normalize(ds.v_feat.float(), dim=1).contiguous(), persistent=False)
        if self._lean:
            pass
        elif custom:
            for m in self._mods:
                if m != "v":
                    self.register_buffer(m + "_feat", F.normalize(getattr(ds, m + "_feat").float(), dim=1).contiguous(),
                                         persistent=False)
        elif not self.is_kwai:
            self.register_buffer("a_feat", F.normalize(ds.a_feat.float(), dim=1).contiguous(), persistent=False)
            if self.dataset_name == "tiktok":
                # :371-378: t_feat is the scatter-mean of word embeddings, built ONCE at init and
                # never recomputed -> a constant, un-normalised [I x 128] table.
                self.word_embedding = nn.Embedding(11574, 128)
                nn.init.xavier_normal_(self.word_embedding.weight)
                self.register_buffer("t_feat", self._word_bag_t_feat(), persistent=False)
            else:
                self.register_buffer("t_feat", F.normalize(ds.t_feat.float(), dim=1).contiguous(), persistent=False)
        for m in self._mods:
            feat = getattr(self, m + "_feat")
            if feat.shape[0] != self.num_items:
                raise ValueError("%s_feat has %d rows, expected num_items=%d" % (m, feat.shape[0], self.num_items))
            if feat.shape[1] % 4 != 0:
                raise ValueError("feature width of '%s' must be a multiple of 4 (got %d)" % (m, feat.shape[1]))
        for m in self._mods:                      # v_dense, a_dense, t_dense in the reference's order (:384-389)
            setattr(self, m + "_dense", nn.Linear(getattr(self, m + "_feat").shape[1], d))
        self.item_feat_dim = d * self.M if self.mm_fusion_mode == "concat" else d
        for m in self._mods:                      # :397-400
            nn.init.xavier_uniform_(getattr(self, m + "_dense").weight)
        self.embedding_user_after_GCN = nn.Linear(self.item_feat_dim, d)
        nn.init.xavier_uniform_(self.embedding_user_after_GCN.weight)
        self.embedding_item_after_GCN = nn.Linear(self.item_feat_dim, d)
        nn.init.xavier_uniform_(self.embedding_item_after_GCN.weight)

    @torch.no_grad()
    def _word_bag_t_feat(self):
        """:371-378: scatter-mean of the word embeddings of every item's words -- [I x 128], not normalised."""
        words = self.dataset.words_tensor
        w = self.word_embedding.weight.detach()
        idx0, idx1 = words[0].to(w.device), words[1].to(w.device)
        rows = int(words[0].max()) + 1
        tot = torch.zeros(rows, 128, device=w.device).index_add_(0, idx0, w[idx1])
        cnt = torch.zeros(rows, device=w.device).index_add_(0, idx0, torch.ones(words.shape[1], device=w.device)).clamp_(min=1)
        return (tot / cnt[:, None]).contiguous()

    def load_state_dict(self, state_dict, strict=True, **kw):
        """nn.Module.load_state_dict + what depends on the loaded tensors without being a checkpoint key: on the tiktok
        data set t_feat is built from word_embedding (:371-378) and the folded constant S_t from t_feat, so both are
        rebuilt from the LOADED word_embedding -- a resume under another seed scores with the features its weights were
        trained with."""
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._plugin.params_changed()
        if self.dataset_name == "tiktok" and hasattr(self, "word_embedding") and "feature_modalities" not in self.config:
            self.t_feat.copy_(self._word_bag_t_feat().to(self.t_feat.device))
            if self._ws is not None and self._ws.get("fold") is not None:
                self._fold_constants(self._ws)
                self._regions = {}            # recorded launches hold the old constants' addresses
                eng = self.__dict__.get("_slab_engine")
                if eng is not None and getattr(eng, "_on_constants_changed", None):
                    eng._on_constants_changed()
        return out

    # ------------------------------------------------------------------ device workspace
    def _device(self):
        return self.embedding_user_after_GCN.weight.device

    def _apply(self, fn, *args, **kwargs):
        """Lean tables: .to(device) / .cuda() move everything EXCEPT the two embedding tables, which stay host parameters
        (state_dict keys as ever); the engine uploads the column slice its rank owns."""
        if not self.__dict__.get("_lean"):
            return super()._apply(fn, *args, **kwargs)
        keep = {k: self._modules[k] for k in ("embedding_user", "embedding_item")}
        order = list(self._modules.keys())
        for k in keep:
            del self._modules[k]
        try:
            return super()._apply(fn, *args, **kwargs)
        finally:
            mods = dict(self._modules)
            mods.update(keep)
            self._modules.clear()
            for k in order:
                self._modules[k] = mods[k]

    def _require_gpu(self):
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("EliMRec (elimrec_amd): the model is on '%s'. The hot path exists only as HIP kernels "
                               "for MI355X; move the model to a GPU (`.to('cuda:0')`). There is no CPU fallback." % dev)
        return dev

    def _workspace(self, B, bwd_rows=None, parity=0):
        """Device buffers for batch size B. bwd_rows = number of gradient rows the backward pass
        will be fed (3*B locally; 3*B*world_size when row gradients are all-gathered). parity: the column-shard engine keeps TWO
        sets of the batch-dependent buffers per batch size and alternates them step by step, so that the planner of step
        t + 1 never writes what a kernel of step t still reads (shard.py: cs_plan)."""
        dev = self._require_gpu()
        bwd_rows = 3 * int(B) if bwd_rows is None else int(bwd_rows)
        parity = int(parity)
        if (self._ws is not None and self._ws_key[:2] == (str(dev), int(B)) and self._ws_key[2] >= bwd_rows
                and self._ws_key[3] == parity):
            return self._ws
        # the batch-dependent buffers of every batch size seen so far are kept (an epoch ends with a ragged batch: B ->
        # tail -> B would otherwise reallocate twice per epoch and invalidate every recorded region)
        sets = self.__dict__.setdefault("_ws_sets", {})
        hit = sets.get((str(dev), int(B), parity))
        if hit is not None and self._ws is not None and self._ws_key[0] == str(dev) and hit[0] >= bwd_rows:
            self._ws.update(hit[2])
            self._ws_key, self._ws_gen = (str(dev), int(B), hit[0], parity), hit[1]
            return self._ws
        key = (str(dev), int(B), bwd_rows, parity)
        self._ws_new = getattr(self, "_ws_new", 0) + 1           # a set allocated (and zero-filled on the CURRENT stream) just now
        N, C, Cy, d = self.num_users + self.num_items, self.C, self.Cy, self.latent_dim
        f32 = dict(dtype=torch.float32, device=dev)
        ws = self._ws if (self._ws is not None and self._ws_key[0] == key[0]) else {}
        self._ws_gen_counter = getattr(self, "_ws_gen_counter", 0) + 1
        self._ws_gen = self._ws_gen_counter                  # recorded regions hold these buffers' addresses
        if "flat_param" not in ws and self._lean:
            # lean tables: the projection weights only (the embeddings are host parameters; the engine holds its column slice
            # of them, of their gradient and of their moments), no table of N rows at all
            self._flatten_parameters(ws)
            ws["fold"] = None
            ws["Y"] = None
        elif "flat_param" not in ws:                        # everything that does not depend on the batch size
            self._flatten_parameters(ws)
            names = ("Out",) if self._folded else (("X0", "Out") if self._bipartite else ("X0", "T0", "T1", "Out", "G"))
            for name in names:
                ws[name] = torch.empty(N, C, **f32)
            if self._wide:
                # the engine's wide form: its own fold (hop kernels), its own tables; here only the parameter views it loads from
                ws["fold"] = None
                ws["X0d"] = ws["flat_param"][:N * d].view(N, d)
                ws["gX0d"] = ws["flat_grad"][:N * d].view(N, d)
            if self._bipartite:
                if not self._folded:
                    ws["H"] = torch.empty(N, d, **f32)
                    ws["bip_ws"] = torch.empty(ops.bipartite_workspace(self.num_users, self.num_items, d, self.M),
                                               dtype=torch.uint8, device=dev)
                if self._folded:
                    ws["Narrow"] = torch.empty(N, d, **f32)        # the part of Out every table shares
                    ws["SrcA"] = torch.empty(N, d, **f32)          # adjoint source tables (active rows only)
                    ws["SrcB"] = torch.empty(N, d, **f32)
                    ws["fold_ws"] = torch.empty(ops.folded_workspace(N, d), dtype=torch.uint8, device=dev)
                    if self.__dict__.get("_skip_fold"):    # the column-sharded engine folds the constants itself, row-sharded
                        ws["fold"] = None
                    else:
                        self._fold_constants(ws)
                    # [E_u ; E_i] and its gradient as ONE [N x d] table: the two embeddings are the first two
                    # tensors of the flat parameter / gradient buffers, back to back
                    eu, ei = self.embedding_user.weight, self.embedding_item.weight
                    gv = ws["grad_views"]
                    adjacent = (ei.data_ptr() == eu.data_ptr() + eu.numel() * 4 and
                                gv["embedding_item.weight"].data_ptr() == gv["embedding_user.weight"].data_ptr() + eu.numel() * 4)
                    if not adjacent:
                        raise RuntimeError("folded propagation expects embedding_user/embedding_item adjacent in the flat buffers")
                    ws["X0d"] = ws["flat_param"][:N * d].view(N, d)
                    ws["gX0d"] = ws["flat_grad"][:N * d].view(N, d)
                else:
                    ws["gXI"] = torch.empty(self.num_items, C, **f32)
            ws["Y"] = torch.zeros(N, Cy, **f32)
            shapes = [(self.num_items, d, getattr(self, m + "_feat").shape[1]) for m in self._mods]
            ws["bwd_w_items"] = torch.empty(max(ops.linear_bwd_w_batched_workspace(shapes), 1), dtype=torch.uint8,
                                            device=dev)
        n3 = bwd_rows
        ws["loss_rows"] = torch.empty(B, **f32)
        ws["grad_rows"] = torch.empty(3 * B, Cy, **f32)
        ws["keys"] = torch.empty(3 * B, dtype=torch.int32, device=dev)          # node id of every local triplet slot
        ws["keys_scratch"] = torch.empty(3 * B, dtype=torch.int32, device=dev)
        ws["slot_seg"] = torch.empty(n3, dtype=torch.int32, device=dev)        # active-row index of every (global) slot
        if self._lazy:
            ws["OutAct"] = torch.empty(n3, C, **f32)               # Out / Y rows of the batch's active nodes
            ws["YAct"] = torch.empty(n3, Cy, **f32)
        ws["active_rows"] = torch.empty(n3, dtype=torch.int32, device=dev)
        ws["dY"] = torch.empty(n3, Cy, **f32)
        ws["seg_info"] = torch.zeros(8, dtype=torch.int32, device=dev)
        ws["plan_ws"] = torch.empty(max(ops.segment_plan_workspace(n3), 1), dtype=torch.uint8, device=dev)
        shapes = [(n3, d, C), (n3, d, C)] + [(n3, d, d)] * self.S
        if self._folded:    # the folded feature projections' weight gradients ride in the same launch
            ws["dOutR"] = torch.empty(n3, C, **f32)                # dLoss/dOut rows in slot order
            shapes += [(n3, d, getattr(self, m + "_feat").shape[1]) for m in self._mods]
        ws["bwd_w_rows"] = torch.empty(max(ops.linear_bwd_w_batched_workspace(shapes), 1), dtype=torch.uint8, device=dev)
        self._ws, self._ws_key = ws, key
        batch_keys_ = ("loss_rows", "grad_rows", "keys", "keys_scratch", "slot_seg", "OutAct", "YAct", "active_rows", "dY",
                       "seg_info", "plan_ws", "dOutR", "bwd_w_rows")
        self.__dict__.setdefault("_ws_sets", {})[(key[0], key[1], parity)] = (bwd_rows, self._ws_gen, {k: ws[k] for k in batch_keys_ if k in ws})
        return ws

    @torch.no_grad()
    def _fold_constants(self, ws):
        """S_m = mean_k A^k [0 ; F_m]  ([N x D_m]) and c = mean_k A^k [0 ; 1]  ([N]), computed once per device
        with the same propagation kernels (narrow part = 0, one table of width D_m)."""
        U, I, L, dev = self.num_users, self.num_items, self.n_layers, self._device()
        P, Q = self._csr("bipP"), self._csr("bipQ")
        fold = {}
        widths = [getattr(self, m + "_feat").shape[1] for m in self._mods] + [4]
        wmax = max(widths)
        for csr in (P, Q):
            csr.build_split(max(wmax, self.C))              # the row-split scratch must cover the widest table
        tmp_ws = torch.empty(ops.bipartite_workspace(U, I, wmax, 1), dtype=torch.uint8, device=dev)
        zeros_u = torch.zeros(U, wmax, dtype=torch.float32, device=dev)
        for m in self._mods:
            feat = getattr(self, m + "_feat")
            D = feat.shape[1]
            out = torch.empty(U + I, D, dtype=torch.float32, device=dev)
            ops.propagate_bipartite(P, Q, U, I, D, 1, L, zeros_u[:, :D].contiguous(), feat, out, tmp_ws)
            fold[m] = out
        ones = torch.ones(I, 4, dtype=torch.float32, device=dev)
        cpad = torch.empty(U + I, 4, dtype=torch.float32, device=dev)
        ops.propagate_bipartite(P, Q, U, I, 4, 1, L, zeros_u[:, :4].contiguous(), ones, cpad, tmp_ws)
        cpad[:, 1:] = 0.0                                   # column 0 = c; the rest pads the bias-gradient GEMM
        fold["c_pad"] = cpad
        fold["c"] = cpad[:, 0].contiguous()
        ws["fold"] = fold

    def _flatten_parameters(self, ws):
        """Re-point every parameter into ONE contiguous fp32 buffer (same Parameter objects, so an
        optimizer created earlier stays valid) and lay the gradients out the same way: the dense Adam
        update over all 18 tensors then becomes a single launch (optim.FusedAdam merges adjacent
        tensors)."""
        params = list(self.named_parameters())
        if self._lean:
            params = [(n_, p) for n_, p in params if not n_.startswith(("embedding_user.", "embedding_item."))]
        dev = self._device()
        sizes = [(p.numel() + 3) // 4 * 4 for _, p in params]          # keep every tensor 16-B aligned
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        flat_grad = torch.zeros_like(flat)
        views, off = {}, 0
        with torch.no_grad():
            for (name, p), n in zip(params, sizes):
                dst = flat[off:off + p.numel()].view_as(p)
                dst.copy_(p.data)
                p.data = dst
                views[name] = flat_grad[off:off + p.numel()].view_as(p)
                off += n
        ws["flat_param"], ws["flat_grad"], ws["grad_views"] = flat, flat_grad, views
        offs, o = {}, 0
        for (name, p), n in zip(params, sizes):
            offs[name] = (o, p.numel())
            o += n
        ws["param_off"] = offs                   # name -> (offset in the flat buffers, numel)
        # every non-embedding parameter, live and as the copy a training forward takes (predict() after an
        # optimizer step must still see the weights its cached tables were computed with)
        tail = 0 if self._lean else sizes[0] + sizes[1]
        ws["tail_off"] = tail
        ws["snap"] = torch.zeros(flat.numel() - tail, dtype=torch.float32, device=dev)
        live, snap, off = {}, {}, 0
        for (name, p), n in zip(params, sizes):
            if off >= tail:
                live[name] = p.data
                snap[name] = ws["snap"][off - tail:off - tail + p.numel()].view_as(p)
            off += n
        ws["live_views"], ws["snap_views"] = live, snap

    def _fusion_weights(self, W=None):
        """[d x C] fusion weights as the kernels consume them. 'mean' fusion (mean over the M
        tables, then a [d x d] Linear; :224-225) is the same map as a [d x C] Linear whose M
        column blocks are all W/M."""
        if W is None:
            wu, wi = self.embedding_user_after_GCN.weight, self.embedding_item_after_GCN.weight
        else:
            wu, wi = W["embedding_user_after_GCN.weight"], W["embedding_item_after_GCN.weight"]
        if self.mm_fusion_mode == "concat":
            return wu, wi
        return (wu.detach() / self.M).repeat(1, self.M).contiguous(), (wi.detach() / self.M).repeat(1, self.M).contiguous()

    def _block_weights(self):
        """Per head block loss weights: fused head 1; single-modal heads alpha if the modality is
        active (:125-126,133-142)."""
        modality = "v" if self.is_kwai else self.modality
        key = (self.predict_type, modality)
        hit = self.__dict__.get("_bw_cache")
        if hit is not None and hit[0] == key:
            return list(hit[1])
        alpha = float(self.config["alpha"])
        w = [1.0]
        for m in self._mods:
            on = (self.predict_type != "normal") and (m in modality)
            w.append(alpha * modality.count(m) if on else 0.0)
        self._bw_cache = (key, tuple(w))
        return w

    # ------------------------------------------------------------------ forward / backward (HIP)
    def _fold_problems(self, ws, W, out, rows=None, rng=None):
        """Out_m = S_m W_m^T + c b_m^T + Narrow for every feature table m -- over all rows into `out`, or gathered
        at `rows` (int32 node ids; slots `rng` = device (begin, end)) into the compact `out`."""
        d, fold = self.latent_dim, ws["fold"]
        fr = self.__dict__.get("_fold_rows") if self.__dict__.get("_slab_fwd") else None
        if rows is not None and fr is not None:
            # the column-sharded engine looked the constants' rows up for this batch (row-sharded / 16-bit storage,
            # lookup.py): compact [R x D_m] rows in active-row order, c and the shared part compact too -- row r, not rows[r]
            n = out.shape[0]
            return [(fr["S"][m][:n], W[m + "_dense.weight"], W[m + "_dense.bias"], out[:, (k + 1) * d:(k + 2) * d], fr["c"][:n],
                     fr["narrow"][:n], None, rng) for k, m in enumerate(self._mods)]
        extra = () if rows is None else (rows, rng)
        return [(fold[m], W[m + "_dense.weight"], W[m + "_dense.bias"], out[:, (k + 1) * d:(k + 2) * d], fold["c"],
                 ws["Narrow"]) + extra for k, m in enumerate(self._mods)]

    @torch.no_grad()
    def _full_tables(self, ws, W):
        """Folded mode: everything after the graph, over all N rows, with the projection weights W."""
        ops.linear_fwd_batched(self._fold_problems(ws, W, ws["Out"]))
        self._head_forward(ws, W)

    @torch.no_grad()
    def _ensure_tables(self):
        """Batch-row mode: the full cached tables of the last training forward, built when somebody reads them."""
        if self._tables_dirty:
            self._tables_dirty = False
            self.__dict__["_slab_engine"].materialize_tables(self._ws)

    @torch.no_grad()
    def _compute_tables(self, ws):
        """compute() + gcn_cf() (:228-272,144-153): fills ws['Out'] and ws['Y'] over all rows."""
        U, I, d, M, C = self.num_users, self.num_items, self.latent_dim, self.M, self.C
        Out, Y = ws["Out"], ws["Y"]
        X0 = ws.get("X0")
        if self._folded:
            # id table + shared user part through the graph at d columns; feature blocks from the folded constants
            self._timed(lambda: ops.propagate_folded(self._csr("adj"), U, I, d, self.n_layers, ws["X0d"], Out[:, :d],
                                                     ws["Narrow"], ws["fold_ws"]))
            self._tables_dirty = False
            self._full_tables(ws, ws["live_views"])
            return
        if self._bipartite:
            ops.copy_cols(self.embedding_item.weight, X0[U:, :d])          # XI block 0 = item id table
        else:
            ops.assemble_x0(self.embedding_user.weight, self.embedding_item.weight, X0, M)
        ops.linear_fwd_batched([(getattr(self, m + "_feat"), getattr(self, m + "_dense").weight,
                                 getattr(self, m + "_dense").bias, X0[U:, (k + 1) * d:(k + 2) * d])
                                for k, m in enumerate(self._mods)])
        if self._bipartite:
            self._timed(lambda: ops.propagate_bipartite(self._csr("bipP"), self._csr("bipQ"), U, I, d, M, self.n_layers,
                                                        self.embedding_user.weight, X0[U:], Out, ws["bip_ws"]))
        else:
            self._propagate(self._csr("adj"), X0, ws["T0"], ws["T1"], Out)
        self._head_forward(ws)

    def _head_forward(self, ws, W=None):
        U, d = self.num_users, self.latent_dim
        Out, Y = ws["Out"], ws["Y"]
        W = ws["live_views"] if W is None else W
        wu, wi = self._fusion_weights(W)
        head = [(Out[:U], wu, W["embedding_user_after_GCN.bias"], Y[:U, :d]),
                (Out[U:], wi, W["embedding_item_after_GCN.bias"], Y[U:, :d])]
        for h, m in enumerate(self._mods):
            blk = slice((h + 1) * d, (h + 2) * d)
            head.append((Out[:, blk], W["s_dense_%s.weight" % m], W["s_dense_%s.bias" % m], Y[:, blk]))
        ops.linear_fwd_batched(head)
        self._publish_cache(Y, dirty=self._tables_dirty)

    def _propagate(self, csr, X0, t0, t1, out):
        self._timed(lambda: ops.propagate(csr, X0, self.n_layers, t0, t1, out))

    def _timed(self, fn, hops=None):
        """Run `hops` (default L) full propagation hops; optionally bracketed by HIP events on the launch stream
        (bench.py)."""
        prof = getattr(self, "_kernel_events", None)
        if prof is None:
            fn()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        prof.append((e0, e1, self.n_layers if hops is None else hops))

    def _publish_cache(self, Y, dirty=False):
        """The tables of this forward are the ones predict() / all_users / all_items / all_s_embs read (views into Y, built
        on first use: a training step only notes where they are)."""
        self._table_version = getattr(self, "_table_version", 0) + 1
        self._cache_src = Y
        self._cache = True
        self._tables_dirty = dirty
        if dirty:
            self._eval_shard = None          # an item-sharded scorer of an older forward (shard.py) is stale now

    # ------------------------------------------------------------------ recorded regions
    def _region(self, name, key, fn):
        """Run fn() -- a fixed sequence of launches on workspace buffers. The first run records the C-ABI calls it
        makes (function + converted arguments); later runs with the same `key` re-issue that list without the
        Python that built it: the step is some 25 launches of 5-50 us each, and issued one by one through the
        tensor-level wrappers the host side costs as much as the GPU side. `key` names everything the launches
        depend on besides buffer contents (sizes, pointers of caller-owned tensors, mode flags, the stream).
        (Capturing the regions as hipGraphs instead was measured in round 2 -- + 12 us per graph launch on this stack -- and
        removed in round 4; a whole step is one host call through csrc/program.hip, elimrec_amd/program.py.)"""
        if self.mm_fusion_mode != "concat" or not self._use_replay:   # 'mean' fusion mixes torch ops into the regions
            return fn()
        key = key + (ops._stream(),)
        name = "%s@%d" % (name, self._ws_gen)       # one entry per batch-size buffer set
        ent = self._regions.get(name)
        if ent is None or ent[0] != key:
            ent = self._regions[name] = [key, 0, None, None]
        if ent[2] is not None:
            _lib.replay(ent[2], name)
            return ent[3]
        calls = []
        _lib.record(calls)
        try:
            out = fn()
        finally:
            _lib.record(None)
        ent[2], ent[3] = calls, out
        return out

    def _fwd_head(self, ws, all_keys, n, B, rank, grad_rows):
        """Everything after the graph at the batch's active rows, generic shapes (the column-shard engine's head when its
        fused 16-row kernels do not cover the shape; the layer means are in ws['OutAct'][:, :d] / the compact shared part
        already): feature blocks, fused Linear and single-modal heads, loss rows and their gradient rows. One recorded region."""
        U, I, d, L = self.num_users, self.num_items, self.latent_dim, self.n_layers
        act, seg = ws["active_rows"][:n], ws["seg_info"]
        bw = self._last_block_weights

        def head():
            OutAct, YAct = ws["OutAct"][:n], ws["YAct"][:n]
            W = ws["live_views"]
            ops.linear_fwd_batched(self._fold_problems(ws, W, OutAct, act, seg[6:8]))
            wu, wi = self._fusion_weights(W)
            bu, bi = W["embedding_user_after_GCN.bias"], W["embedding_item_after_GCN.bias"]
            none3 = (None, None, None)
            problems = [(OutAct, wu, bu, YAct[:, :d]) + none3 + (seg[2:4],), (OutAct, wi, bi, YAct[:, :d]) + none3 + (seg[4:6],)]
            for h, m in enumerate(self._mods):
                blk = slice((h + 1) * d, (h + 2) * d)
                problems.append((OutAct[:, blk], W["s_dense_%s.weight" % m], W["s_dense_%s.bias" % m], YAct[:, blk]) + none3
                                + (seg[6:8],))
            ops.linear_fwd_batched(problems)
            ops.bpr_head_rows(YAct, ws["slot_seg"][3 * B * rank:3 * B * (rank + 1)], d, bw, ws["loss_rows"], grad_rows)

        self._region("fwd_head", (self._ws_gen, all_keys.data_ptr(), n, B, rank, tuple(bw), grad_rows is not None), head)
        self._publish_cache(ws["Y"], dirty=True)

    def _index_tensors(self, *ts):
        """int64, contiguous, on the model's device (no-ops for tensors that already are)."""
        dev = self._device()
        return tuple(t if (t.dtype == torch.int64 and t.device == dev and t.is_contiguous())
                     else t.to(device=dev, dtype=torch.int64).contiguous() for t in ts)

    @torch.no_grad()
    def batch_keys(self, users, pos, neg):
        """int32 [3B] node ids of the triplet slots (3b: user, 3b+1: U + pos, 3b+2: U + neg) -- the rows the loss
        reads (getEmbedding's gathers, :274-289) and the keys of the head-gradient rows."""
        B = int(users.numel())
        ws = self._workspace(B, getattr(self, "_bwd_rows_hint", None))
        users, pos, neg = self._index_tensors(users, pos, neg)
        return ops.triplet_rows(users, pos, neg, self.num_users, ws["keys"][:3 * B], I=self.num_items, err=self._index_err())

    def _index_err(self):
        e = self.__dict__.get("_index_err_word")
        if e is None or e.device != self._device():
            e = self.__dict__["_index_err_word"] = torch.zeros(1, dtype=torch.int32, device=self._device())
        return e

    def check_indices(self):
        """Raise IndexError if any batch since the last call held a user / item id outside the tables (the reference
        fails at the gather, models/EliMRec.py:277-281; here the kernels flag it, keep running in bounds, and the host
        looks at the flag when it synchronises anyway -- once per epoch in main.py)."""
        e = self.__dict__.get("_index_err_word")
        if e is None:
            return
        bits = int(e.item())
        if bits:
            e.zero_()
            what = [n for b, n in enumerate(("user", "positive item", "negative item")) if bits >> b & 1]
            raise IndexError("EliMRec: %s index out of range in a training batch" % " / ".join(what))

    @torch.no_grad()
    def _forward_hip(self, users, pos, neg, need_grad, all_keys=None, rank=0):
        """Forward on this rank's triplets. all_keys: the node ids of every rank's slots in rank order (default:
        this rank's own). The segment plan of those keys (unique active nodes, slot -> active row) is made FIRST:
        it depends on the indices only, the forward evaluates the head at the active rows, and the backward
        reduces the gathered gradient rows with it."""
        B = int(users.numel())
        self._fwd_gen = getattr(self, "_fwd_gen", 0) + 1
        users, pos, neg = self._index_tensors(users, pos, neg)
        keys = self.batch_keys(users, pos, neg)
        ws = self._ws
        all_keys = keys if all_keys is None else all_keys
        n = int(all_keys.numel())
        if n > ws["slot_seg"].numel():
            raise RuntimeError("workspace holds %d gradient rows, the gathered batch has %d" % (ws["slot_seg"].numel(), n))
        self._plan_n = n
        self._last_block_weights = self._block_weights()
        grad_rows = ws["grad_rows"] if need_grad else None
        ops.segment_plan(all_keys, self.num_users, self.num_users + self.num_items, ws["active_rows"][:n], ws["seg_info"],
                         ws["slot_seg"][:n], ws["plan_ws"])
        self._compute_tables(ws)
        ops.bpr_head(ws["Y"], self.num_users, self.num_items, users, pos, neg, self.latent_dim,
                     self._last_block_weights, ws["loss_rows"], grad_rows, ws["keys_scratch"] if need_grad else None)
        loss = torch.empty((), dtype=torch.float32, device=self._device())
        ops.fixed_order_sum(ws["loss_rows"], loss)
        return loss

    @torch.no_grad()
    def _backward_batch_rows(self, ws, gscale, grad_rows, n, head_only=True, pack_bwd=None, merge=None,
                             defer_reduce=False, sources=None):
        """The head's backward at the batch's active rows (the column-shard engine's cs_backward_local): gradient rows ->
        active rows (the forward's plan) -> dOut rows, head and projection weight gradients from the compact Out / dY rows.
        Two recorded regions; the adjoint propagation is the engine's."""
        U, I, d, M, C, S = self.num_users, self.num_items, self.latent_dim, self.M, self.C, self.S
        dY, seg, act = ws["dY"][:n], ws["seg_info"], ws["active_rows"][:n]
        dOutR, OutAct, gv, fold = ws["dOutR"][:n], ws["OutAct"][:n], ws["grad_views"], ws["fold"]
        bw = self._last_block_weights
        heads_on = [h for h in range(S) if bw[1 + h] != 0.0]     # heads switched off carry an all-zero gradient block
        concat = self.mm_fusion_mode == "concat"
        f32 = dict(dtype=torch.float32, device=self._device())
        grads = {}

        def head_input():
            wu, wi = self._fusion_weights()
            head_ws = [getattr(self, "s_dense_" + m).weight for m in self._mods]
            ops.segment_apply_head_bwd(grad_rows, act, seg, dY, ws["plan_ws"], U, d, C, [h + 1 for h in range(S)], wu, wi,
                                       head_ws, dOutR, scale=gscale, pack_bwd=pack_bwd, sources=sources)

        def head_weights():
            # fusion Linears: dW = dY_f^T . Out[active rows], user slots / item slots separately
            problems, fused_tmp = [], {}
            for name, rng in (("embedding_user_after_GCN", seg[2:4]), ("embedding_item_after_GCN", seg[4:6])):
                gw = gv[name + ".weight"] if concat else torch.empty(d, C, **f32)
                fused_tmp[name] = gw
                problems.append(dict(A=dY[:, :d], B=OutAct, out=gw, rng=rng, colsum=gv[name + ".bias"]))
                grads[name + ".weight"], grads[name + ".bias"] = gv[name + ".weight"], gv[name + ".bias"]
            for h in heads_on:
                name, blk = "s_dense_" + self._mods[h], slice((h + 1) * d, (h + 2) * d)
                problems.append(dict(A=dY[:, blk], B=OutAct[:, blk], out=gv[name + ".weight"], rng=seg[6:8],
                                     colsum=gv[name + ".bias"]))
                grads[name + ".weight"], grads[name + ".bias"] = gv[name + ".weight"], gv[name + ".bias"]
            # feature projections: Out_m = S_m W_m^T + c b_m^T + (shared part)  =>  dW_m = dOut_m^T S_m over the
            # active rows only, db_m = dOut_m^T c
            fr = self.__dict__.get("_fold_rows") if self.__dict__.get("_slab_fwd") else None
            for k, m in enumerate(self._mods):
                if fr is not None:      # the constants' rows of this batch, compact (looked up by the engine): row r itself
                    problems.append(dict(A=dOutR[:, (k + 1) * d:(k + 2) * d], B=fr["S"][m][:n], out=gv[m + "_dense.weight"],
                                         rng=seg[6:8], colsum=gv[m + "_dense.bias"], colsum_weight=fr["c"]))
                else:
                    problems.append(dict(A=dOutR[:, (k + 1) * d:(k + 2) * d], B=fold[m], out=gv[m + "_dense.weight"],
                                         row_index=act, rng=seg[6:8], colsum=gv[m + "_dense.bias"], colsum_weight=fold["c"]))
                grads[m + "_dense.weight"], grads[m + "_dense.bias"] = gv[m + "_dense.weight"], gv[m + "_dense.bias"]
            handle = ops.linear_bwd_w_batched(problems, ws["bwd_w_rows"], merge=merge, defer_reduce=bool(defer_reduce) and concat,
                                              defer_all=defer_reduce == "all" and concat)
            if not concat:  # 'mean' fusion: fold the M replicated column blocks back into the [d x d] weight
                for name, gw in fused_tmp.items():
                    gv[name + ".weight"].copy_(gw.view(d, M, d).sum(1) / M)
            return dict(grads), handle

        key = (self._ws_gen, grad_rows.data_ptr(), gscale.data_ptr(), n, tuple(bw))
        self._region("bwd_head_in", key + (0 if pack_bwd is None else pack_bwd.data_ptr(),
                                           0 if sources is None else (sources[1] if sources[0] == "split" else sources[0].data).data_ptr()),
                     head_input)
        mkey = (str(defer_reduce),) + (() if merge is None else (merge["rows"].data_ptr(), merge["keys"].data_ptr(),
                                                                 merge["mask"].data_ptr()))
        grads, self._bwd_w_reduce = self._region("bwd_head_w", key + mkey, head_weights)
        return dict(grads)

    @torch.no_grad()
    def _backward_hip(self, gscale, grad_rows=None):
        """Returns {parameter name: gradient tensor}. gscale: device fp32[1] (d loss_total / d loss).
        grad_rows default to the rows the last local forward produced; a data-parallel driver passes the rows
        gathered from every rank (in the order of the keys the forward planned)."""
        ws = self._ws
        dev = self._device()
        U, I, d, M, C, S = self.num_users, self.num_items, self.latent_dim, self.M, self.C, self.S
        grad_rows = ws["grad_rows"] if grad_rows is None else grad_rows
        n_rows = grad_rows.shape[0]
        if n_rows != self._plan_n:
            raise RuntimeError("backward got %d gradient rows, the forward planned %d" % (n_rows, self._plan_n))
        dY, seg, act = ws["dY"][:n_rows], ws["seg_info"], ws["active_rows"][:n_rows]
        ops.segment_apply(grad_rows, seg, dY, ws["plan_ws"], scale=gscale)
        bw = self._last_block_weights
        heads_on = [h for h in range(S) if bw[1 + h] != 0.0]
        wu, wi = self._fusion_weights()
        G0 = ws.get("X0")
        if not self._bipartite:
            G0.zero_()      # the bipartite path masks inactive rows instead of reading zeros
        # heads switched off by the modality ablation carry an all-zero gradient block
        head_ws = [getattr(self, "s_dense_" + m).weight for m in self._mods]
        if self._folded:    # dLoss/dOut rows of the active nodes, in slot order
            dOutR = ws["dOutR"][:n_rows]
            ops.head_bwd_input(dY, act, seg, U, d, C, [h + 1 for h in range(S)], wu, wi, head_ws, 1.0, None, compact=dOutR)
        else:
            ops.head_bwd_input(dY, act, seg, U, d, C, [h + 1 for h in range(S)], wu, wi, head_ws, 1.0, G0)
        grads = {}
        f32 = dict(dtype=torch.float32, device=dev)
        out_rows, out_index = ws["Out"], act
        # fusion Linears: dW = dY_f^T . Out[active rows], user slots / item slots separately
        gv = ws["grad_views"]
        concat = self.mm_fusion_mode == "concat"
        problems, fused_tmp = [], {}
        for name, rng in (("embedding_user_after_GCN", seg[2:4]), ("embedding_item_after_GCN", seg[4:6])):
            gw = gv[name + ".weight"] if concat else torch.empty(d, C, **f32)
            fused_tmp[name] = gw
            problems.append(dict(A=dY[:, :d], B=out_rows, out=gw, row_index=out_index, rng=rng, colsum=gv[name + ".bias"]))
            grads[name + ".weight"], grads[name + ".bias"] = gv[name + ".weight"], gv[name + ".bias"]
        for h in heads_on:
            name = "s_dense_" + self._mods[h]
            problems.append(dict(A=dY[:, (h + 1) * d:(h + 2) * d], B=out_rows[:, (h + 1) * d:(h + 2) * d],
                                 out=gv[name + ".weight"], row_index=out_index, rng=seg[6:8], colsum=gv[name + ".bias"]))
            grads[name + ".weight"], grads[name + ".bias"] = gv[name + ".weight"], gv[name + ".bias"]
        if self._folded:
            # feature projections: Out_m = S_m W_m^T + c b_m^T + (shared part)  =>  dW_m = dOut_m^T S_m over the
            # active rows only, db_m = dOut_m^T c  (the dense I-row contraction of the unfolded path disappears)
            fold = ws["fold"]
            for k, m in enumerate(self._mods):
                problems.append(dict(A=dOutR[:, (k + 1) * d:(k + 2) * d], B=fold[m], out=gv[m + "_dense.weight"],
                                     row_index=act, rng=seg[6:8], colsum=gv[m + "_dense.bias"], colsum_weight=fold["c"]))
                grads[m + "_dense.weight"], grads[m + "_dense.bias"] = gv[m + "_dense.weight"], gv[m + "_dense.bias"]
        ops.linear_bwd_w_batched(problems, ws["bwd_w_rows"])
        if not concat:      # 'mean' fusion: fold the M replicated column blocks back into the [d x d] weight
            for name, gw in fused_tmp.items():
                gv[name + ".weight"].copy_(gw.view(d, M, d).sum(1) / M)
        # back through the propagation (A^T; A itself when symmetric), then the layer-0 pieces
        gu, gi = gv["embedding_user.weight"], gv["embedding_item.weight"]
        if self._folded:
            AT = self._csr("adj" if self._adj_symmetric else "adjT")
            self._timed(lambda: ops.propagate_folded_bwd(AT, U, I, d, M, self.n_layers, dOutR, act, seg, ws["SrcA"],
                                                         ws["SrcB"], ws["gX0d"], ws["fold_ws"]))
            grads["embedding_user.weight"], grads["embedding_item.weight"] = gu, gi
            return grads
        if self._bipartite:
            ops.blocksum_rows(G0, act, seg, d, M, ws["H"])
            sym = self._adj_symmetric
            PT, QT = self._csr("bipQ" if sym else "bipPT"), self._csr("bipP" if sym else "bipQT")
            gXI = ws["gXI"]
            self._timed(lambda: ops.propagate_bipartite_bwd(PT, QT, U, I, d, M, self.n_layers, G0, ws["H"], act, seg,
                                                            gXI, gu, ws["bip_ws"]))
            ops.copy_cols(gXI[:, :d], gi)
            g_items = gXI
        else:
            G = ws["G"]
            self._propagate(self._csr("adj" if self._adj_symmetric else "adjT"), G0, ws["T0"], ws["T1"], G)
            ops.embed_grad(G, U, I, d, M, gu, gi)
            g_items = G[U:]
        grads["embedding_user.weight"], grads["embedding_item.weight"] = gu, gi
        problems = []
        for k, m in enumerate(self._mods):
            problems.append(dict(A=g_items[:, (k + 1) * d:(k + 2) * d], B=getattr(self, m + "_feat"),
                                 out=gv[m + "_dense.weight"], colsum=gv[m + "_dense.bias"]))
            grads[m + "_dense.weight"], grads[m + "_dense.bias"] = gv[m + "_dense.weight"], gv[m + "_dense.bias"]
        ops.linear_bwd_w_batched(problems, ws["bwd_w_items"])
        return grads

    # ------------------------------------------------------------------ engine API (elimrec_amd/dist.py)
    def forward_local(self, users, pos, neg, all_keys=None, rank=0, world_size=1):
        """Forward on this rank's triplets; all_keys = batch_keys() of every rank, concatenated in rank order.
        Returns (loss, grad_rows [3B x Cy]): d(local mean loss)/dY rows of this rank's slots."""
        B = int(users.numel())
        if self._lazy:
            raise RuntimeError("forward_local / backward_global (elimrec_amd.dist.DataParallelTrainer) drive the unfolded row-major "
                               "forms only (--propagation=full|bipartite, --head_rows=all, layer_num < 2); this model trains on the "
                               "column-shard engine: model.bpr_loss -> backward -> FusedAdam.step, or ColumnShardTrainer.step")
        self._bwd_rows_hint = 3 * B * world_size
        self._workspace(B, 3 * B * world_size)
        loss = self._forward_hip(users, pos, neg, need_grad=True, all_keys=all_keys, rank=rank)
        return loss, self._ws["grad_rows"]

    def backward_global(self, grad_rows, scale):
        """Backward from the gradient rows of every rank (rank order); `scale` is a device fp32[1]."""
        return self._backward_hip(scale, grad_rows=grad_rows)

    # ------------------------------------------------------------------ reference API
    def bpr_loss(self, users, pos_items, neg_items):
        """models/EliMRec.py:115-142. int64 index tensors of shape [b] -> 0-dim loss supporting
        .backward(retain_graph=True) and .cpu().item() (main.py:98-102)."""
        self._require_gpu()
        if self.is_kwai:
            self.modality = "v"                        # :133-134
        if self._lazy:
            # the column-shard engine's step, run when the caller's optimizer steps (plugin.py): loss.backward() and
            # FusedAdam.step() complete the request; whatever is read in between is made real first
            return self._plugin.begin(users, pos_items, neg_items)
        return _BprLossFn.apply(self, users, pos_items, neg_items, *self._all_params())

    @property
    def plugin(self):
        """The controller behind bpr_loss -> backward -> optimizer.step (plugin.py): `.engine`, `.trainer` once a step ran."""
        return self._plugin

    def state_dict(self, *args, **kwargs):
        """nn.Module.state_dict with the embedding tables written back from the engine's master copy first."""
        self._plugin.sync_params()
        return super().state_dict(*args, **kwargs)

    def _all_params(self):
        params = self.__dict__.get("_param_list")
        if params is None:      # the Parameter objects are fixed after construction (they are re-pointed, never replaced)
            params = self.__dict__["_param_list"] = [p for _, p in self.named_parameters()]
        return params

    def _no_lean(self, what):
        if self.__dict__.get("_wide"):
            raise RuntimeError("--propagation=folded on an adjacency with a diagonal: %s is not available (the folded form of such "
                               "an adjacency exists on the column-sharded engine only); train with ColumnShardTrainer" % what)
        if self.__dict__.get("_lean"):
            raise RuntimeError("--lean_tables=1: %s is not available (no table of N rows lives outside the column-sharded "
                               "engine); train with ColumnShardTrainer, evaluate with evaluate() / predict()" % what)

    def compute(self):
        """:228-272. Returns (all_users [U x d], all_items [I x d]). Under torch.no_grad(): views into Y. With autograd
        enabled: differentiable copies (_TablesFn) -- what getEmbedding / the generic BasicModel losses build on."""
        self._require_gpu()
        self._no_lean("compute()")
        self._plugin.settle()
        self._plugin.sync_params()               # the table build below reads the parameters themselves
        if torch.is_grad_enabled():
            self._last_tables = _TablesFn.apply(self, *self._all_params())
            return self._last_tables
        self._slab_fwd = False
        ws = self._workspace(self._ws_key[1] if self._ws_key else 1, parity=self._ws_key[3] if self._ws_key else 0)
        self._compute_tables(ws)
        self._last_tables = (self.all_users, self.all_items)
        return self._last_tables

    @torch.no_grad()
    def _backward_tables(self, g_users, g_items):
        """HIP backward from dense gradients of all_users / all_items: every node is an active row whose head-gradient
        row holds its table gradient in the fused block and zeros in the single-modal blocks."""
        U, I, d, Cy = self.num_users, self.num_items, self.latent_dim, self.Cy
        N = U + I
        dev = self._device()
        ws = self._workspace(self._ws_key[1] if self._ws_key else 1, N, parity=self._ws_key[3] if self._ws_key else 0)
        rows = torch.zeros(N, Cy, dtype=torch.float32, device=dev)
        if g_users is not None:
            rows[:U, :d] = g_users
        if g_items is not None:
            rows[U:, :d] = g_items
        keys = torch.arange(N, dtype=torch.int32, device=dev)
        ops.segment_plan(keys, U, N, ws["active_rows"][:N], ws["seg_info"], ws["slot_seg"][:N], ws["plan_ws"])
        self._plan_n = N
        self._last_block_weights = [1.0] + [0.0] * self.S
        lazy, self._lazy = self._lazy, False          # the tables were built over all rows: the full-row backward
        try:
            return self._backward_hip(torch.ones(1, dtype=torch.float32, device=dev), grad_rows=rows)
        finally:
            self._lazy = lazy

    def gcn_cf(self, detach=False):
        """:144-153. The single-modal head tables computed by the last compute()."""
        return self.all_s_embs

    def getEmbedding(self, users, pos_items, neg_items):
        """:274-289: rows of the fused tables for users / positives / negatives + the raw ("ego") embedding rows. The
        table rows carry autograd (through compute()); the ego rows are returned detached -- no loss of the reference
        reads them (BasicModel.py:59-113 only unpacks them)."""
        all_users, all_items = self.compute()
        ego_u, ego_i = self.embedding_user.weight.detach(), self.embedding_item.weight.detach()
        neg = (all_items[neg_items], ego_i[neg_items]) if neg_items is not None else (None, None)
        return all_users[users], all_items[pos_items], neg[0], ego_u[users], ego_i[pos_items], neg[1]

    def forward(self, users, items):
        """:299-307."""
        all_users, all_items = self.compute()
        return torch.sum(all_users[users] * all_items[items], dim=1).detach()

    def _head_mask(self):
        modality = "v" if self.is_kwai else self.modality
        return sum(1 << h for h, m in enumerate(self._mods) if m in modality)

    @torch.no_grad()
    def predict_device(self, user_ids, scores=None, top_k=0, train_ptr=None, train_items=None, tie_order="id"):
        """Device-side predict (+ optional train-item masking and top-K). Uses the tables cached
        by the LAST training forward, like the reference (:98-99). tie_order: the lists' order among equal
        scores -- "id" (lowest item id first) or "reference" (evaluate.h:26-33's partial_sort_copy, replayed on the device)."""
        dev = self._require_gpu()
        self._plugin.realise_forward()
        if self._ws is None or self._cache is None:
            raise RuntimeError("predict() needs the tables cached by a training forward (call bpr_loss or compute first)")
        self._ensure_tables()
        users = torch.as_tensor(user_ids, device=dev).long().contiguous()
        B, I = users.numel(), self.num_items
        sh = self.__dict__.get("_eval_shard")
        if sh is not None:       # several ranks with row-sharded tables: this rank scores its block of the items (shard_eval.py)
            if scores is not None:
                scores.copy_(sh.scores(users, train_ptr, train_items))
            if top_k:
                if tie_order != "id":
                    raise ValueError("item-sharded tables rank by (score, id); the evaluator re-ranks tied rows in the reference's order")
                return sh.topk(users, top_k, train_ptr, train_items)
            return None, None
        # top-K only (the evaluator): no [B x I] score block in the workspace, the catalogue is scored in chunks
        import os
        chunked = scores is None and top_k > 0
        need = ops.score_workspace(B, self.num_users, I, self.S, max(top_k, 1), topk_only=chunked, d=self.latent_dim)
        if self._ws.get("score_ws") is None or self._ws["score_ws"].numel() < need:
            self._ws["score_ws"] = torch.empty(need, dtype=torch.uint8, device=dev)
        idx = val = None
        if top_k:
            idx = torch.empty(B, top_k, dtype=torch.int32, device=dev)
            val = torch.empty(B, top_k, dtype=torch.float32, device=dev)
        # block norms of the cached tables: computed once per table version, reused by every user block
        if self._ws.get("sqn_version") != self._table_version:
            if self._ws.get("sqn") is None:
                self._ws["sqn"] = torch.empty(self.num_users + I, 1 + self.S, dtype=torch.float32, device=dev)
            ops.row_sqnorms(self._ws["Y"], self.latent_dim, 1 + self.S, self._ws["sqn"])
            self._ws["sqn_version"] = self._table_version
        ops.score_topk(self._ws["Y"], self.num_users, I, users, self.latent_dim, self.S, self._head_mask(),
                       self.fusion_mode, self.predict_type, self._ws["score_ws"], scores=scores, K=top_k,
                       topk_idx=idx, topk_val=val, train_ptr=train_ptr, train_items=train_items, sqnorm=self._ws["sqn"],
                       tie_order=tie_order)
        return idx, val

    def predict(self, user_ids, candidate_items=None):
        """:96-113. CPU fp32 tensor [len(user_ids) x I]; `candidate_items` is ignored as in the reference."""
        dev = self._require_gpu()
        scores = torch.empty(len(user_ids), self.num_items, dtype=torch.float32, device=dev)
        self.predict_device(user_ids, scores=scores)
        return scores.cpu()
