"""``SGConv`` / ``GCNConv`` modules shaped like the torch_geometric operators the
reference instantiates (model/SGC.py:11,14; model/GCN.py:9-16), computed with the
same CSR SpMM kernels as the SGNN path.

torch_geometric is third-party, un-vendored and version-unpinned in the
reference (README.md:34); the semantics implemented are the documented ones:
remaining self-loops with weight 1, degree summed at the target index,
``deg^-1/2`` with inf -> 0, ``out[target] += dinv[src] * w * dinv[target] * x[src]``.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .norm_adj import NormAdj


class _Propagate(torch.autograd.Function):
    """``A_hat^K x`` with ``(A_hat^T)^K grad`` as its backward."""

    @staticmethod
    def forward(ctx, x, adj, K):
        ctx.adj, ctx.K = adj, K
        return adj.propagate(x.detach(), K)

    @staticmethod
    def backward(ctx, grad):
        g = None
        if ctx.needs_input_grad[0]:
            g = ctx.adj.transpose().propagate(grad.contiguous(), ctx.K)
        return g, None, None


def propagate(x, adj, K=1):
    if x.requires_grad and torch.is_grad_enabled():
        return _Propagate.apply(x, adj, K)
    return adj.propagate(x.detach(), K)


class _GraphConvBase(nn.Module):
    def __init__(self, add_self_loops=True, cached=False):
        super().__init__()
        self.add_self_loops = add_self_loops
        self.cached = cached
        self._adj = None
        self._adj_src = None

    def _norm_adj(self, x, edge_index, edge_weight):
        # the normalised adjacency depends only on the graph: rebuild it when the
        # edge tensors change, not on every forward (torch_geometric with
        # cached=False redoes gcn_norm per call; the result is the same).  The key is
        # the tensor objects themselves, kept alive here so that their storage cannot be
        # freed and handed to a different graph, plus their in-place version counters.
        src = self._adj_src
        same = (self._adj is not None and src is not None and src[0] is edge_index and src[1] is edge_weight
                and src[2] == edge_index._version
                and src[3] == (None if edge_weight is None else edge_weight._version) and src[4] == x.shape[0])
        if not same:
            self._adj = NormAdj.from_edge_index(edge_index, x.shape[0], device=x.device,
                                                edge_weight=edge_weight, add_self_loops=self.add_self_loops)
            self._adj_src = (edge_index, edge_weight, edge_index._version,
                             None if edge_weight is None else edge_weight._version, x.shape[0])
        return self._adj


class SGConv(_GraphConvBase):
    """``lin(A_hat^K x)`` — torch_geometric.nn.SGConv as used at model/SGC.py:11.

    With ``cached=False`` (torch_geometric's default, and what model/SGC.py:11 builds) the
    propagation is redone on every forward.  ``A_hat^K (x W^T) = (A_hat^K x) W^T``, so when
    the layer narrows the features (602 -> 41 on Reddit) and nothing is cached the linear
    map is applied first and the K hops run at the output width — the same result to fp32
    rounding at a fraction of the gather traffic (``reorder=False`` keeps the literal
    order).  With ``cached=True`` the propagated features are kept at the input width and
    only the linear map runs per forward, as in torch_geometric."""

    def __init__(self, in_channels, out_channels, K=1, cached=False, add_self_loops=True, bias=True,
                 reorder=True):
        super().__init__(add_self_loops, cached)
        self.in_channels, self.out_channels, self.K = in_channels, out_channels, K
        self.reorder = reorder
        self.lin = nn.Linear(in_channels, out_channels, bias=bias)
        self._cached_x = None

    def reset_parameters(self):
        self.lin.reset_parameters()
        self._cached_x = None

    def forward(self, x, edge_index, edge_weight=None):
        if self.cached and self._cached_x is not None:
            return self.lin(self._cached_x)
        adj = self._norm_adj(x, edge_index, edge_weight)
        if not self.cached and self.reorder and self.out_channels < self.in_channels:
            out = propagate(torch.nn.functional.linear(x, self.lin.weight), adj, self.K)
            return out if self.lin.bias is None else out + self.lin.bias
        h = propagate(x, adj, self.K)
        if self.cached:
            self._cached_x = h.detach()
        return self.lin(h)


class GCNConv(_GraphConvBase):
    """``A_hat (x W^T) + b`` — torch_geometric.nn.GCNConv as used at model/GCN.py:9-10."""

    def __init__(self, in_channels, out_channels, cached=False, add_self_loops=True, bias=True):
        super().__init__(add_self_loops, cached)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        bound = math.sqrt(6.0 / (self.in_channels + self.out_channels))   # glorot, as torch_geometric
        nn.init.uniform_(self.lin.weight, -bound, bound)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, x, edge_index, edge_weight=None):
        h = self.lin(x)
        out = propagate(h, self._norm_adj(x, edge_index, edge_weight), 1)
        if self.bias is not None:
            out = out + self.bias
        return out
