"""`MLP(input_dim, dimensions, activation, dropout)` with the reference's interface (util/mlp.py:6-38): a chain of
Linears with activation + dropout between layers and none after the last, `init_weight('xavier' | 'normal')`.

The live EliMRec model does not import it (its projections are bare Linears, models/EliMRec.py:88-90,384-407); it is the
per-modality projection helper the predecessors used, kept so that a model written against the reference's util
package finds it. Every Linear runs in csrc/gemm.hip (fp32-input MFMA): forward with the activation fused into the
epilogue for 'relu' (other activations are applied by their torch op on the device), backward through the same
library (input gradient = a Linear with the transposed weight, weight gradient = elimrec_linear_bwd_w). Widths that
are not multiples of 4 are zero-padded (exact: the padding multiplies zeros)."""
import torch
import torch.nn.functional as F
from torch import nn

from . import ops


def _pad4(t):
    """Zero-pad the last dimension to a multiple of 4 (contiguous copy only when needed)."""
    k = t.shape[-1]
    if k % 4 == 0 and t.is_contiguous():
        return t
    return F.pad(t, (0, (-k) % 4)).contiguous()


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        if not x.is_cuda:
            raise RuntimeError("elimrec_amd.MLP runs on the GPU only (input is on '%s')" % x.device)
        x2 = _pad4(x.reshape(-1, x.shape[-1]).float())
        w = _pad4(weight.float())
        out = torch.empty(x2.shape[0], w.shape[0], dtype=torch.float32, device=x.device)
        ops.linear_fwd(x2, w, None if bias is None else bias.float().contiguous(), out, act="relu" if relu else None)
        ctx.save_for_backward(x2, w, out if relu else None)
        ctx.relu, ctx.in_features, ctx.lead, ctx.has_bias = relu, weight.shape[1], x.shape[:-1], bias is not None
        return out.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, w, out = ctx.saved_tensors
        g = gy.reshape(-1, gy.shape[-1]).float()
        if ctx.relu:
            g = g * (out > 0)
        g = _pad4(g)
        n_out, k = w.shape
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wt = _pad4(w.t())                                     # [k x n_out(+pad)]: dX = g . W
            gx = torch.empty(g.shape[0], k, dtype=torch.float32, device=g.device)
            ops.linear_fwd(g, wt, None, gx)
            gx = gx[:, :ctx.in_features].reshape(*ctx.lead, ctx.in_features)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            n1 = g.shape[1]
            gwp = torch.empty(n1, k, dtype=torch.float32, device=g.device)
            gbp = torch.empty(n1, dtype=torch.float32, device=g.device)
            ws = torch.empty(max(ops.linear_bwd_w_workspace(g.shape[0], n1, k), 1), dtype=torch.uint8, device=g.device)
            ops.linear_bwd_w(g, x2, gwp, ws, colsum=gbp)
            gw = gwp[:n_out, :ctx.in_features]
            gb = gbp[:n_out] if ctx.has_bias else None
        return gx, gw, gb, None


class MLP(nn.Module):
    def __init__(self, input_dim, dimensions, activation="relu", dropout=0.):
        super(MLP, self).__init__()
        self.input_dim = input_dim
        self.dimensions = dimensions
        self.activation = activation
        self.dropout = dropout
        self.linears = nn.ModuleList([nn.Linear(input_dim, dimensions[0])])
        for din, dout in zip(dimensions[:-1], dimensions[1:]):
            self.linears.append(nn.Linear(din, dout))

    def init_weight(self, t="xavier"):
        for lin in self.linears:
            if t == "xavier":
                nn.init.xavier_uniform_(lin.weight)
            elif t == "normal":
                nn.init.normal_(lin.weight, std=0.1)

    def forward(self, x):
        last = len(self.linears) - 1
        for i, lin in enumerate(self.linears):
            fused = i < last and self.activation == "relu"
            x = _LinearFn.apply(x, lin.weight, lin.bias, fused)
            if i < last:
                if not fused:
                    x = F.__dict__[self.activation](x)
                if self.dropout > 0:
                    x = F.dropout(x, self.dropout, training=self.training)
        return x
