"""torch.optim.Adam semantics (coupled L2, main.py:49) with the update running in csrc/optim.hip."""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    """Drop-in for `optim.Adam(params, lr, weight_decay)`; parameters without a gradient are
    skipped (no decay, no moment update, step count untouched), as torch does.

    Launch merging: when consecutive parameters (and their gradients) sit back to back in memory --
    EliMRec lays its parameters and gradients out in one flat buffer each -- and share the same step
    count, they are updated by ONE kernel launch over the contiguous span; the moments are kept in
    flat buffers that mirror the parameter storage so they are contiguous too."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._mirror = {}     # storage data_ptr -> (exp_avg_flat, exp_avg_sq_flat)
        self._plan = None     # (signature, merged runs) of the last step: same gradient tensors -> same launches
        # parameters of an EliMRec model whose training step is deferred to this call (plugin.py): step() runs the engine's
        # whole step when a backward() was requested for the pending loss
        self._ctl = next((p.__dict__["_elimrec_ctl"] for g in self.param_groups for p in g["params"]
                          if p.__dict__.get("_elimrec_ctl") is not None), None)
        if self._ctl is not None:
            self._ctl.register_optimizer(self)

    def zero_grad(self, set_to_none=True):
        ctl = self._ctl
        if ctl is not None and set_to_none and not ctl.grads_set and not ctl.grads_deferred() and not ctl.any_raw_grad():
            return              # no .grad is set (the deferred step keeps its gradients inside the engine)
        super().zero_grad(set_to_none=set_to_none)
        if ctl is not None and set_to_none:
            ctl.grads_set = False

    def _state_for(self, p):
        st = self.state[p]
        if st:
            return st
        st["step"] = 0
        storage = p.untyped_storage()
        nfloat = storage.nbytes() // 4
        if p.is_contiguous() and nfloat > p.numel():
            # p is a view into a larger flat buffer: mirror the whole buffer once
            key = storage.data_ptr()
            if key not in self._mirror:
                self._mirror[key] = (torch.zeros(nfloat, dtype=torch.float32, device=p.device),
                                     torch.zeros(nfloat, dtype=torch.float32, device=p.device))
            m, v = self._mirror[key]
            off = p.storage_offset()
            st["exp_avg"] = m[off:off + p.numel()].view_as(p)
            st["exp_avg_sq"] = v[off:off + p.numel()].view_as(p)
        else:
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        ctl = self._ctl
        if ctl is not None:
            if ctl.step(self):
                return loss
            if ctl.engine is not None:
                ctl.sync_params()           # the update below reads and writes the embedding parameters themselves: current
                ctl.params_changed()        # values first, and the master copy reloads from them before the next step
        # fast path: the same gradient buffers as last step (a trainer that keeps its gradients in fixed memory)
        # => the same merged launches, one step further
        sig = tuple(0 if p.grad is None else p.grad.data_ptr() for group in self.param_groups for p in group["params"])
        if self._plan is not None and self._plan[0] == sig:
            for r, group, states in self._plan[1]:
                r["step"] += 1
                for st in states:
                    st["step"] = r["step"]
                ops.adam_step_raw(r["p"], r["g"], r["m"], r["v"], r["n"], group["lr"], group["betas"][0], group["betas"][1],
                                  group["eps"], group["weight_decay"], r["step"])
            return loss
        plan = []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            run = None      # [p_ptr, g_ptr, m_ptr, v_ptr, numel, step, tensors...]
            def flush(r):
                if r is not None:
                    ops.adam_step_raw(r["p"], r["g"], r["m"], r["v"], r["n"], group["lr"], b1, b2, group["eps"],
                                      group["weight_decay"], r["step"])
                    plan.append((r, group, r.pop("states")))
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam: parameters must be on a HIP device (no CPU fallback)")
                st = self._state_for(p)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous")
                ptrs = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr())
                n = p.numel()
                if run is not None and run["step"] == st["step"] and all(
                        0 <= ptrs[i] - (run[k] + 4 * run["n"]) <= 12 and (ptrs[i] - run[k]) % 4 == 0
                        and ptrs[i] - (run[k] + 4 * run["n"]) == ptrs[0] - (run["p"] + 4 * run["n"])
                        for i, k in enumerate(("p", "g", "m", "v"))):
                    run["n"] = (ptrs[0] - run["p"]) // 4 + n      # swallow the (zero) alignment padding
                    run["keep"].append(g)
                    run["states"].append(st)
                else:
                    flush(run)
                    run = dict(p=ptrs[0], g=ptrs[1], m=ptrs[2], v=ptrs[3], n=n, step=st["step"], keep=[g], states=[st],
                               contiguous=p.grad.is_contiguous())
            flush(run)
        # the plan is reusable only if no gradient had to be copied to a temporary
        self._plan = (sig, plan) if all(r["contiguous"] for r, _, _ in plan) else None
        return loss

    # ------------------------------------------------------------------ checkpoint / resume
    @torch.no_grad()
    def export_state(self, named_params):
        """{parameter name: {"step", "exp_avg", "exp_avg_sq"}} on the CPU for the parameters that have state."""
        out = {}
        for name, p in named_params:
            st = self.state.get(p)
            if st:
                out[name] = dict(step=int(st["step"]), exp_avg=st["exp_avg"].detach().cpu().clone(),
                                 exp_avg_sq=st["exp_avg_sq"].detach().cpu().clone())
        return out

    @torch.no_grad()
    def import_state(self, named_params, saved):
        """Inverse of export_state: the moments are copied INTO this optimizer's own (flat, mirrored) buffers, so the
        merged single-launch update keeps working after a resume."""
        for name, p in named_params:
            if name in saved:
                st = self._state_for(p)
                st["step"] = int(saved[name]["step"])
                st["exp_avg"].copy_(saved[name]["exp_avg"].to(p.device))
                st["exp_avg_sq"].copy_(saved[name]["exp_avg_sq"].to(p.device))
        self._plan = None
