"""torch.ops.elimrec.* -- the TORCH_LIBRARY registration of the C ABI (csrc/torch_ops.cpp).

    from elimrec_amd import torch_ops
    ops = torch_ops.load()                      # torch.ops.elimrec
    out = ops.propagate(rowptr, col, val, X, 3)

The ops run on HIP tensors only (dispatch key CUDA = HIP on ROCm); there is no CPU kernel behind them, so a CPU tensor
raises NotImplementedError from the dispatcher. The package itself keeps calling the library through ctypes
(_lib.py): its step replays recorded argument lists, which the dispatcher cannot do; both bindings drive the same
libelimrec_hip.so.
"""
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libelimrec_torch.so")
_loaded = False


def load():
    global _loaded
    if not _loaded:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("elimrec_amd: %s is missing. Build it with `make -C elimrec_amd/csrc` (or "
                               "__graft_entry__.build())." % LIB_PATH)
        torch.ops.load_library(LIB_PATH)
        _loaded = True
    return torch.ops.elimrec


OPS = ("propagate", "linear_fwd", "linear_bwd_w", "bpr_head_fwd", "adam_step_", "score_topk", "rank_metrics", "sample_triplets")
