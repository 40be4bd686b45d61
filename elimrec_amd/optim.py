"""torch.optim.Adam semantics (coupled L2, main.py:49) with the update running in csrc/optim.hip."""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    """Drop-in for `optim.Adam(params, lr, weight_decay)`; parameters without a gradient are
    skipped (no decay, no moment update, step count untouched), as torch does."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam: parameters must be on a HIP device (no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adam_step(p.data, g, st["exp_avg"], st["exp_avg_sq"], group["lr"], b1, b2, group["eps"],
                              group["weight_decay"], st["step"])
        return loss
