"""elimrec_amd -- the EliMRec per-batch hot path (models/EliMRec.py of Xiaohao-Liu/EliMRec)
as hand-written gfx950 HIP kernels behind the reference's own plugin surface."""
from .configurator import Configurator
from .data_iterator import DataIterator
from .dataset import Dataset, SyntheticDataset, csr_to_user_dict
from .logger import Logger, Meter
from .basic_model import BasicModel
from .model import EliMRec
from .mlp import MLP
from .optim import FusedAdam
from .sampler import PairwiseSamplerV2
from .shard import ColumnShardEngine, ColumnShardTrainer
from .evaluator import ProxyEvaluator, UniEvaluator
from . import capacity


def set_seed(seed):
    """util/tool.py:11-16."""
    import numpy as np
    import torch
    np.random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    torch.manual_seed(seed)
