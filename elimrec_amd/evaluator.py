"""Full-catalogue top-K evaluation with the reference's facade (evaluator/proxy_evaluator.py:40-108,
evaluator/backend/cpp/uni_evaluator.py:37-203): `ProxyEvaluator(...).evaluate(model)` ->
(float32 ndarray over metrics x top_show, tab-joined "%.8f" string).

What changed underneath: for a model that exposes `predict_device`, scoring, train-item masking,
top-K and the metric curves all run on the GPU (csrc/eval.hip) and only the per-user metric rows
([users x metrics*K] floats) come back to the host for the final mean -- the reference instead
copies a [128 x I] score matrix to the host per batch and ranks it on 8 CPU threads.

Tie rule: the reference's std::partial_sort_copy breaks ties in the C++ library's heap order
(SURVEY quirk 6). tie_order = "reference" (the default) gives exactly those lists: the library
algorithm is replayed on the device, one wave per user, inside the scoring call (csrc/eval.hip
ref_order_kernel) -- same cost as tie_order = "id", the device's own rule (score descending, item
id ascending). The two agree on every row without tied scores at or across the K boundary.
"""
import numpy as np
import torch

from . import ops
from .data_iterator import DataIterator

metric_dict = {"Precision": 1, "Recall": 2, "MAP": 3, "NDCG": 4, "MRR": 5}
re_metric_dict = {v: k for k, v in metric_dict.items()}


class UniEvaluator(object):
    def __init__(self, dataset, user_train_dict, user_test_dict, user_neg_test=None, metric=None, top_k=50,
                 batch_size=1024, num_thread=8):
        if not isinstance(user_train_dict, dict):
            raise TypeError("user_train_dict must be a dict")
        if user_test_dict is not None and not isinstance(user_test_dict, dict):
            raise TypeError("user_test_dict must be a dict or None")
        if metric is None:
            metric = ["Precision", "Recall", "MAP", "NDCG", "MRR"]
        elif isinstance(metric, str):
            metric = [metric]
        elif not isinstance(metric, (set, tuple, list)):
            raise TypeError("The type of 'metric' (%s) is invalid!" % metric.__class__.__name__)
        for m in metric:
            if m not in metric_dict:
                raise ValueError("There is not the metric named '%s'!" % metric)
        if user_neg_test is not None:
            raise NotImplementedError("sampled-negative evaluation (rec.evaluate.neg > 0) is not on the EliMRec path")
        self.dataset = dataset
        self.user_pos_train = user_train_dict
        self.user_pos_test = user_test_dict
        self.user_neg_test = user_neg_test
        self.metrics_num = len(metric)
        self.metrics = [metric_dict[m] for m in metric]
        self.num_thread = num_thread          # kept for interface parity; ranking runs on the GPU
        self.batch_size = batch_size
        # users scored per launch. The reference's test_batch_size (128) bounds the [batch x I] host matrix it ranks on the
        # CPU; on the device the per-user results do not depend on the grouping, and a larger block lets eight workgroups
        # share every item tile through L2 and amortises the launches and the users' operand loads of every catalogue chunk
        # (128: 0.048 s, 1024: 0.036 s, 2048: 0.0335 s per validation pass at the Tiktok shape with EXACT math; with the default
        # math 2048: 0.0271, 4096: 0.0258, 8192: 0.0253, 32768: 0.0250 -- 8192 users x 16384 items is a 537 MB score block)
        self.block_users = max(int(batch_size), 8192)
        self.workspace_gib = 8.0                # the scorer's workspace budget: users per launch are halved until it fits
        self.max_top = top_k if isinstance(top_k, int) else max(top_k)
        self.top_show = np.arange(top_k) + 1 if isinstance(top_k, int) else np.sort(top_k)
        self._dev_cache = {}
        self._default_users = None
        # ties among equal scores: "reference" (default) = the reference's lists (evaluate.h:26-33's partial_sort_copy replayed on
        # the device, every row); "id" = the device's own rule (score descending, item id ascending) (--tie_order)
        self.tie_order = "reference"
        self.tie_rows_replayed = 0
        # the default scorer (dot products as six bf16 piece products, csrc/eval.hip score_t16b_kernel) returned a wrong score once
        # in round 3 on one device and never again in 49 000 replays: every evaluation re-scores its first
        # users with the fp32-MFMA scorer and compares the K returned scores -- a difference beyond the two forms' round-off is
        # counted, logged, and the rest of the evaluation runs on the fp32 scorer (scorer_check_users = 0: no check)
        self.scorer_check_users = 1024
        # ... in the first evaluation of a run and in every 16th after it: the check re-scores 1024 users with the slower scorer,
        # 0.8 ms of a 13 ms pass at the Tiktok shape
        self.scorer_check_every = 16
        self._evaluations = 0
        self.scorer_checked_rows = self.scorer_mismatch_rows = 0
        self.range_violations = 0          # scorer waves that saw a score outside their launch's range invariant, over all passes

    def _cross_check_scorer(self, model, users, cache_key=None):
        """Top-K of the first `scorer_check_users` users by the default (bf16 x 3) scorer and by the fp32-MFMA scorer, compared ON THE
        DEVICE: the number of rows whose returned scores differ by more than 1e-6 (the forms agree to 2.4e-7) stays a device scalar,
        which _cross_check_verdict reads once the pass's own launches are enqueued -- the check costs its two scorer calls, not a host
        round trip in front of the pass. Returns (count tensor, users checked) or None when there is nothing to check."""
        from . import _lib
        lib = _lib.load()
        if (not users or self.scorer_check_users <= 0 or int(lib.elimrec_score_get_math()) == 0 or int(lib.elimrec_score_get_bf16x3()) == 0
                or model.latent_dim not in (32, 64) or getattr(model, "_eval_shard", None) is not None
                or self.max_top > min(128, model.num_items)):
            return None
        device = model._require_gpu()
        key = (str(device), "check", cache_key, self.scorer_check_users) if cache_key is not None else None
        hit = self._dev_cache.get(key) if key is not None else None
        if hit is None:
            users = list(users[:self.scorer_check_users])
            train_ptr, train_items = self._batch_csr(users, self.user_pos_train, device, unique=False)
            users_t = torch.as_tensor(np.asarray(users, dtype=np.int64)).to(device)
            hit = (users_t, train_ptr, train_items)
            if key is not None:
                self._dev_cache[key] = hit
        users_t, train_ptr, train_items = hit
        _, val_a = model.predict_device(users_t, top_k=self.max_top, train_ptr=train_ptr, train_items=train_items)
        lib.elimrec_score_set_bf16x3(0)
        try:
            _, val_b = model.predict_device(users_t, top_k=self.max_top, train_ptr=train_ptr, train_items=train_items)
        finally:
            lib.elimrec_score_set_bf16x3(1)
        diff = (val_a - val_b).abs()
        bad = (torch.where(torch.isfinite(val_a) & torch.isfinite(val_b), diff, (val_a != val_b).float()) > 1e-6).any(1).sum()
        return bad, int(users_t.numel())

    def _range_verdict(self, model, first, act=True):
        """Reads (and clears) the scorers' range-invariant counter behind the pass's launches. Violations under the default
        (bf16 x 3) scorer switch the process to the fp32 scorer and the pass is scored again; under the fp32 scorer they are an
        error: the cached tables themselves must hold non-finite values. act = False (several ranks: a rank must not leave the
        collectives of the pass on its own): counted and logged only."""
        from . import _lib
        n = ops.score_range_violations(reset=True)
        self.range_violations += n
        if not n:
            return 0
        if not act:
            from .logger import Logger
            Logger.info("[evaluator] %d scorer waves of this rank saw a score outside the range of predict type %s / fusion %s"
                        % (n, model.predict_type, model.fusion_mode))
            return 0
        lib = _lib.load()
        from .logger import Logger
        if first and int(lib.elimrec_score_get_bf16x3()) and int(lib.elimrec_score_get_math()) == 1 and model.latent_dim in (32, 64):
            Logger.info("[evaluator] %d scorer waves saw a score outside the range of predict type %s / fusion %s: the fp32 scorer is "
                        "used from here on, this pass is scored again" % (n, model.predict_type, model.fusion_mode))
            lib.elimrec_score_set_bf16x3(0)
            return n
        raise FloatingPointError("the evaluator's scores left the range of predict type %s / fusion %s in %d scorer waves (fp32 scorer): "
                                 "the cached tables hold non-finite or corrupted values" % (model.predict_type, model.fusion_mode, n))

    def _cross_check_verdict(self, pending):
        """Reads the cross-check's count (a host synchronisation: call it behind the pass's launches). On any differing row the
        process keeps the fp32 scorer from here on; returns the number of such rows."""
        if pending is None:
            return 0
        from . import _lib
        bad, n = int(pending[0]), pending[1]
        self.scorer_checked_rows += n
        self.scorer_mismatch_rows += bad
        if bad:
            from .logger import Logger
            Logger.info("[evaluator] %d of %d cross-checked users got different top-%d scores from the bf16x3 scorer and the fp32-MFMA "
                        "scorer (> 1e-6): the fp32 scorer is used from here on, this pass is scored again" % (bad, n, self.max_top))
            _lib.load().elimrec_score_set_bf16x3(0)
        return bad

    def metrics_info(self):
        cols = ["\t".join(("%s@" % re_metric_dict[m] + str(k)).ljust(12) for k in self.top_show) for m in self.metrics]
        return "metrics:\t%s" % "\t".join(cols)

    def _batch_csr(self, users, table, device, unique):
        lists = [sorted(set(table.get(u, []))) if unique else table.get(u, []) for u in users]
        ptr = np.zeros(len(users) + 1, dtype=np.int64)
        np.cumsum([len(x) for x in lists], out=ptr[1:])
        flat = np.fromiter((i for x in lists for i in x), dtype=np.int32, count=int(ptr[-1]))
        return torch.from_numpy(ptr).to(device), torch.from_numpy(flat).to(device)

    def evaluate(self, model, test_users=None, shard=None):
        """shard = (rank, world): this process scores a contiguous 1/world slice of the users and the per-user metric rows
        are summed across the ranks (all_reduce of the zero-filled [users x metrics*K] matrix), so every rank ends with
        the same rows -- and, the final mean being taken over the same matrix, the same bits -- as a single process.
        Default: the ranks of an initialised torch.distributed job (the cached tables are replicated on every rank)."""
        if test_users is None:
            if getattr(self, "_default_users", None) is None:
                self._default_users = list(self.user_pos_test.keys())
            test_users = self._default_users
        cached = test_users is self._default_users
        if not isinstance(test_users, (list, tuple, set, np.ndarray)):
            raise TypeError("'test_user' must be a list, tuple, set or numpy array!")
        if not hasattr(model, "predict_device"):
            raise TypeError("model must expose predict_device(); host-side ranking is not part of this package")
        test_users = list(test_users)
        all_dev = self.metric_rows(model, test_users, shard=shard, cached=cached)
        all_rows = all_dev.cpu().numpy()                                  # [users, metrics*K]
        final = np.mean(all_rows, axis=0).reshape(self.metrics_num, self.max_top)[:, self.top_show - 1].reshape(-1)
        buf = "\t".join(("%.8f" % x).ljust(12) for x in final)
        return final, buf

    def metric_rows(self, model, test_users, shard=None, cached=False, reduce=True):
        """Per-user metric rows of `test_users` on the device, [len(test_users), metrics*K]; with shard = (rank, world)
        only this rank's slice is computed (the rest zero) and `reduce` sums the matrix over the ranks."""
        import torch.distributed as dist
        if shard is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            shard = (dist.get_rank(), dist.get_world_size())
        n = len(test_users)
        sharded = shard is not None and shard[1] > 1
        if hasattr(model, "_ensure_tables") and getattr(model, "_cache", None) is not None and not sharded:
            model._ensure_tables()       # (lean tables on one rank: the item-shard scorer exists once the tables are built)
        if sharded and hasattr(model, "_ensure_tables"):
            # every rank, before any rank can run out of users: materialising the cached tables of a multi-rank job is a
            # collective (ColumnShardEngine.materialize_tables), and a rank with an empty slice would otherwise skip it
            model._ensure_tables()
        if getattr(model, "_eval_shard", None) is not None:
            # the cached tables are ITEM-sharded (row-sharded constants, shard_eval.py): every rank scores every user
            # against its items and the merged lists -- hence the metric rows -- are identical on all ranks
            sharded = False
        lo, hi = (n * shard[0] // shard[1], n * (shard[0] + 1) // shard[1]) if sharded else (0, n)
        alloc = torch.zeros if sharded else torch.empty
        all_dev = alloc(n, self.metrics_num * self.max_top, dtype=torch.float32, device=model._require_gpu())
        at = lo
        mine = test_users[lo:hi]
        block = self._users_per_launch(model)
        check = self._evaluations % self.scorer_check_every == 0
        self._evaluations += 1
        pending = self._cross_check_scorer(model, mine, cache_key=(lo, hi) if cached else None) if check else None
        for attempt in (0, 1):
            at = lo
            for k, batch_users in enumerate(DataIterator(mine, batch_size=block, shuffle=False, drop_last=False)):
                key = (k, lo, hi, block) if cached else None
                self.evaluate_batch(model, batch_users, cache_key=key, out=all_dev[at:at + len(batch_users)])
                at += len(batch_users)
            # the scorer's cross-check of this pass's first users, read behind the pass's launches; a mismatch switches the process to
            # the fp32 scorer and the pass is scored once more with it. Likewise the range invariant EVERY scorer launch of the pass
            # checked in its epilogue (every score is a sigmoid of a bounded argument: csrc/eval.hip ScoreArgs::lo / hi)
            out_of_range = self._range_verdict(model, first=attempt == 0, act=not sharded and getattr(model, "_eval_shard", None) is None)
            if attempt == 1 or not (self._cross_check_verdict(pending) or out_of_range):
                break
        if sharded and reduce:
            dist.all_reduce(all_dev, op=dist.ReduceOp.SUM)
        return all_dev

    def _users_per_launch(self, model):
        """block_users, halved until the scorer's workspace for this catalogue (or this rank's item shard) fits the budget
        (workspace_gib, default 8): a recdim outside the chunked scorer's set needs a [users x items] score block, which
        at 12.5 M items per rank is 50 MB per user."""
        block = self.block_users
        sh = getattr(model, "_eval_shard", None)
        # (item shards differ by one item between ranks when I % W != 0, and the sharded scorer's collectives run once per user
        # block: the block size must come out the same on every rank, so it is sized from the LARGEST shard)
        bounds = getattr(sh, "bounds", None)
        n_items = (max(b - a for a, b in zip(bounds[:-1], bounds[1:])) if bounds else sh.i1 - sh.i0) if sh is not None else model.num_items
        budget = float(self.workspace_gib) * 2 ** 30
        while block > 16 and ops.score_workspace(block, model.num_users, n_items, model.S, self.max_top, topk_only=True,
                                                 d=model.latent_dim) > budget:
            block //= 2
        return block

    def _topk_in_reference_order(self, model, users_t, train_ptr, train_items):
        """tie_order = "reference": the lists evaluate.h:26-33 makes of these users' masked score rows -- std::partial_sort_copy's
        heap order among equal scores -- computed ON THE DEVICE inside the scoring call (csrc/eval.hip ref_order_kernel: one wave
        per user replays the library algorithm operation for operation on the scores the scorer has just produced, the heap carried
        from catalogue chunk to chunk). Every row, tied or not; no score row is materialised or copied, nothing synchronises with
        the host, and the pass costs what the "id" order costs whatever the state of training.
        Item-sharded tables (several ranks, shard_eval.py): a rank sees only its items, and the algorithm is sequential over the
        catalogue -- the merged (score, id) lists are the reference's on every row without a tie at or across K; the rows that
        have one get their whole score rows (all-gathered, as predict() returns them) ranked by the same device kernel."""
        K = self.max_top
        if getattr(model, "_eval_shard", None) is None:
            return model.predict_device(users_t, top_k=K, train_ptr=train_ptr, train_items=train_items, tie_order="reference")
        if K + 1 > min(256, model.num_items):
            raise ValueError("tie_order=reference over item-sharded tables needs the merged top-(K + 1) list: K + 1 = %d exceeds "
                             "min(256, num_items = %d); evaluate with --tie_order=id" % (K + 1, model.num_items))
        idx1, val1 = model.predict_device(users_t, top_k=K + 1, train_ptr=train_ptr, train_items=train_items)
        idx, val = idx1[:, :K].contiguous(), val1[:, :K].contiguous()
        tied = torch.nonzero((val1[:, :-1] == val1[:, 1:]).any(1)).flatten()       # (every rank holds the same merged lists)
        self.tie_rows_replayed += int(tied.numel())
        # <= 1 GiB of score rows at a time (a row of configs[4]'s catalogue is 400 MB); every rank holds the same merged lists, hence
        # the same `tied` and the same number of (collective) score-row calls
        step = max(1, (1 << 28) // max(1, model.num_items))
        for a in range(0, int(tied.numel()), step):
            part = tied[a:a + step]
            lo_t, hi_t = train_ptr[part], train_ptr[part + 1]
            sub_ptr = torch.zeros(part.numel() + 1, dtype=torch.int64, device=users_t.device)
            torch.cumsum(hi_t - lo_t, 0, out=sub_ptr[1:])
            take = torch.repeat_interleave(lo_t - sub_ptr[:-1], hi_t - lo_t) + torch.arange(int(sub_ptr[-1]), device=users_t.device)
            it = train_items[take] if take.numel() else torch.zeros(1, dtype=torch.int32, device=users_t.device)
            sc = torch.empty(part.numel(), model.num_items, dtype=torch.float32, device=users_t.device)
            model.predict_device(users_t[part], scores=sc, train_ptr=sub_ptr, train_items=it.contiguous())
            ti = torch.empty(part.numel(), K, dtype=torch.int32, device=users_t.device)
            tv = torch.empty(part.numel(), K, dtype=torch.float32, device=users_t.device)
            ops.topk_reference_order(sc, K, ti, tv)
            idx[part], val[part] = ti, tv
        return idx, val

    def evaluate_batch(self, model, batch_users, return_topk=False, cache_key=None, out=None):
        """Per-user metric rows [len(batch_users), metrics*K] (device tensor) for one user block.
        cache_key: the user blocks of the default evaluation order are the same every time, so their
        index tensors (user ids, train-mask CSR, truth CSR) are built once and stay on the device."""
        device = model._require_gpu()
        key = (str(device), cache_key) if cache_key is not None else None
        hit = self._dev_cache.get(key) if key is not None else None
        if hit is None:
            train_ptr, train_items = self._batch_csr(batch_users, self.user_pos_train, device, unique=False)
            truth_ptr, truth_items = self._batch_csr(batch_users, self.user_pos_test, device, unique=True)
            users_t = torch.as_tensor(np.asarray(batch_users, dtype=np.int64)).to(device)
            hit = (users_t, train_ptr, train_items, truth_ptr, truth_items)
            if key is not None:
                self._dev_cache[key] = hit
        users_t, train_ptr, train_items, truth_ptr, truth_items = hit
        if self.tie_order == "reference":
            idx, val = self._topk_in_reference_order(model, users_t, train_ptr, train_items)
        else:
            idx, val = model.predict_device(users_t, top_k=self.max_top, train_ptr=train_ptr, train_items=train_items)
        if out is None:
            out = torch.empty(len(batch_users), self.metrics_num * self.max_top, dtype=torch.float32, device=device)
        ops.rank_metrics(idx, truth_ptr, truth_items, self.metrics, out)
        return (out, idx, val) if return_topk else out


class ProxyEvaluator(object):
    def __init__(self, dataset, user_train_dict, user_test_dict, user_neg_test=None, metric=None, group_view=None,
                 top_k=50, batch_size=1024, num_thread=8):
        if group_view is not None:
            raise NotImplementedError("group_view evaluation is out of scope (NeuRec.properties:27 sets None; the "
                                      "reference's GroupedEvaluator cannot be constructed as shipped)")
        self.evaluator = UniEvaluator(dataset, user_train_dict, user_test_dict, user_neg_test, metric=metric,
                                      top_k=top_k, batch_size=batch_size, num_thread=num_thread)

    def metrics_info(self):
        return self.evaluator.metrics_info()

    def evaluate(self, model):
        return self.evaluator.evaluate(model)
