"""elimrec_amd -- the EliMRec per-batch hot path (models/EliMRec.py of Xiaohao-Liu/EliMRec)
as hand-written gfx950 HIP kernels behind the reference's own plugin surface."""
import os as _os

# The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4), and a dependency between two
# streams costs far more when it crosses queues than the kernels around it: a training step has three streams in flight at most
# (compute, the plan / exchange stream, the library's RCCL calls), and measured on MI355X (ms per step, one host call per step):
#   one rank                 1 queue 0.361   2 0.307   3 0.303   4 0.303
#   multi-rank path (1 rank) 1 queue 0.496   2 0.391   3 0.392   4 0.493   8 0.772
# Read by the runtime when it initialises (the first HIP call of the process); an explicit setting of the caller's wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")

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
