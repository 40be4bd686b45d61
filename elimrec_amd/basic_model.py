"""Plugin base class with the reference's method set (models/BasicModel.py:9-113): builds the
valid/test evaluators from the config, names checkpoints, exposes evaluate()/test()."""
import os

import torch
import torch.nn.functional as F
from torch import nn

from .evaluator import ProxyEvaluator


class BasicModel(nn.Module):
    def __init__(self, dataset, config):
        super(BasicModel, self).__init__()
        self.config = config
        self.dataset = dataset
        train = dataset.get_user_train_dict()
        common = dict(metric=config["metric"], group_view=config["group_view"], top_k=config["topks"],
                      batch_size=config["test_batch_size"], num_thread=config["num_thread"])
        self.valid_evaluator = ProxyEvaluator(dataset, train, dataset.get_user_valid_dict(), None, **common)
        self.test_evaluator = ProxyEvaluator(dataset, train, dataset.get_user_test_dict(), None, **common)
        if "tie_order" in config:        # --tie_order=reference (CLI-only): the reference's lists among equal scores too (evaluator.py)
            if str(config["tie_order"]) not in ("id", "reference"):
                raise ValueError("tie_order must be id or reference")
            for ev in (self.valid_evaluator, self.test_evaluator):
                ev.evaluator.tie_order = str(config["tie_order"])
        self.infonce_criterion = nn.CrossEntropyLoss()          # BasicModel.py:32

    def getFileName(self):
        """`{path}/{recommender}-{dataset}-{loss}-{suffix}.pth.tar` (BasicModel.py:34-40)."""
        cfg = self.config
        os.makedirs(cfg["path"], exist_ok=True)          # (several ranks of one job arrive here together)
        name = "%s-%s-%s-%s.pth.tar" % (cfg["recommender"], cfg["data.input.dataset"], cfg["loss"], cfg["suffix"])
        return os.path.join(cfg["path"], name)

    def predict(self, user_ids, candidate_items=None):
        raise NotImplementedError

    def compute(self):
        raise NotImplementedError

    def getEmbedding(self, users, pos_items, neg_items):
        raise NotImplementedError

    def evaluate(self):
        return self.valid_evaluator.evaluate(self)

    def test(self):
        return self.test_evaluator.evaluate(self)

    # ---- generic losses on top of getEmbedding (BasicModel.py:59-113). EliMRec overrides bpr_loss; these are the
    # reference's base-class versions, differentiable through EliMRec.compute()'s autograd bridge.
    def bpr_loss(self, users, pos, neg):
        """BasicModel.py:59-79: softplus(neg - pos) on un-normalised embedding rows."""
        users_emb, pos_emb, neg_emb, _, _, _ = self.getEmbedding(users.long(), pos.long(), neg.long())
        pos_scores = torch.sum(torch.mul(users_emb, pos_emb), dim=1)
        neg_scores = torch.sum(torch.mul(users_emb, neg_emb), dim=1)
        return torch.mean(F.softplus(neg_scores - pos_scores))

    def infonce(self, users, pos, neg=None):
        """BasicModel.py:81-95 (in-batch negatives; `neg` accepted and ignored so the driver's three-argument call works)."""
        users_emb, pos_emb, _, _, _, _ = self.getEmbedding(users.long(), pos.long(), None)
        users_emb = F.normalize(users_emb, dim=1)
        pos_emb = F.normalize(pos_emb, dim=1)
        logits = torch.mm(users_emb, pos_emb.T) / self.temp
        labels = torch.arange(users.shape[0], device=logits.device)
        return self.infonce_criterion(logits, labels)

    def fast_loss(self, users, pos, neg=None):
        """BasicModel.py:97-113."""
        users_emb, pos_emb, _, _, _, _ = self.getEmbedding(users.long(), pos.long(), None)
        alpha = self.config["alpha"]
        users_emb = F.normalize(users_emb, dim=1)
        pos_emb = F.normalize(pos_emb, dim=1)
        all_users, all_items = self._last_tables          # the tables of THIS forward, with their autograd graph
        all_users = F.normalize(all_users, dim=1)
        all_items = F.normalize(all_items, dim=1)
        pos_scores = torch.sum(torch.mul(users_emb, pos_emb), dim=1)
        pos_loss = torch.sum((alpha - 1) * torch.pow(pos_scores, 2) - 2 * alpha * pos_scores)
        all_loss = torch.trace(torch.matmul(torch.matmul(all_users.T, all_users), torch.matmul(all_items.T, all_items)))
        return pos_loss + all_loss
