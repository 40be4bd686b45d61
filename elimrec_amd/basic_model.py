"""Plugin base class with the reference's method set (models/BasicModel.py:9-113): builds the
valid/test evaluators from the config, names checkpoints, exposes evaluate()/test()."""
import os

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

    def getFileName(self):
        """`{path}/{recommender}-{dataset}-{loss}-{suffix}.pth.tar` (BasicModel.py:34-40)."""
        cfg = self.config
        if not os.path.exists(cfg["path"]):
            os.mkdir(cfg["path"])
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

    def bpr_loss(self, users, pos, neg):
        raise NotImplementedError

    def infonce(self, users, pos):
        raise NotImplementedError("infonce (BasicModel.py:81-95) is not on the EliMRec hot path: the reference "
                                  "driver is run with --loss=bpr_loss; see DESIGN.md, out of scope")

    def fast_loss(self, users, pos):
        raise NotImplementedError("fast_loss (BasicModel.py:97-113) is not on the EliMRec hot path; see DESIGN.md")
