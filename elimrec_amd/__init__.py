"""elimrec_amd -- the EliMRec per-batch hot path (models/EliMRec.py of Xiaohao-Liu/EliMRec)
as hand-written gfx950 HIP kernels behind the reference's own plugin surface."""

# (GPU_MAX_HW_QUEUES -- the HIP runtime's cap on hardware queues per process, which decides what a dependency between two of the
# step's streams costs -- is a process-wide knob of the HOST application: main.py and bench.py set it for their own process
# before the runtime initialises (3: compute stream, plan / exchange stream, one spare for copies and RCCL's own work; measured
# ms per step with 1 / 2 / 3 / 4 / 8 queues: one rank 0.361 / 0.307 / 0.303 / 0.303 / -, the multi-rank path on one rank
# 0.496 / 0.391 / 0.392 / 0.493 / 0.772). Importing this package does not touch it.)
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
