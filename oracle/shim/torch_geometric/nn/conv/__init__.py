"""Restated MessagePassing (aggr='add'/'mean', flow source_to_target) + SAGEConv (oracle shim)."""
import inspect
import types
import torch
from torch import nn
from ..dense.linear import Linear


class MessagePassing(nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=0, **kwargs):
        super().__init__()
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim
        self._msg_params = list(inspect.signature(self.message).parameters)

    def propagate(self, edge_index, size=None, **kwargs):
        src, dst = edge_index[0], edge_index[1]
        x = kwargs.get("x")
        x_pair = x if isinstance(x, (tuple, list)) else (x, x)
        N = x_pair[1].shape[0] if (size is None or size[1] is None) else size[1]
        margs = {}
        for p in self._msg_params:
            if p.endswith("_j"):
                v = kwargs[p[:-2]]
                v = v[0] if isinstance(v, (tuple, list)) else v
                margs[p] = v.index_select(0, src)
            elif p.endswith("_i"):
                v = kwargs[p[:-2]]
                v = v[1] if isinstance(v, (tuple, list)) else v
                margs[p] = v.index_select(0, dst)
            elif p == "index":
                margs[p] = dst
            elif p == "ptr":
                margs[p] = None
            elif p == "size_i":
                margs[p] = N
            else:
                margs[p] = kwargs[p]
        msg = self.message(**margs)
        out = torch.zeros((N,) + tuple(msg.shape[1:]), dtype=msg.dtype, device=msg.device)
        out.index_add_(0, dst, msg)
        if self.aggr == "mean":
            cnt = torch.zeros(N, dtype=msg.dtype, device=msg.device).index_add_(
                0, dst, torch.ones_like(dst, dtype=msg.dtype))
            out = out / cnt.clamp(min=1).view(-1, *([1] * (msg.dim() - 1)))
        return out

    def message(self, x_j):
        return x_j


class SAGEConv(MessagePassing):
    """out_i = W_l mean_j x_j + b_l + W_r x_i  (models.py:227-236)."""
    def __init__(self, in_channels, out_channels, normalize=False, root_weight=True, bias=True, **kw):
        super().__init__(aggr="mean")
        self.lin_l = Linear(in_channels, out_channels, bias=bias)
        self.root_weight = root_weight
        if root_weight:
            self.lin_r = Linear(in_channels, out_channels, bias=False)

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        if self.root_weight:
            self.lin_r.reset_parameters()

    def forward(self, x, edge_index):
        out = self.lin_l(self.propagate(edge_index, x=x))
        if self.root_weight:
            out = out + self.lin_r(x)
        return out

    def message(self, x_j):
        return x_j


gat_conv = types.ModuleType("gat_conv")
sage_conv = types.ModuleType("sage_conv")
from . import gcn_conv  # noqa: E402
