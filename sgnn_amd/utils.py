"""Drop-in replacements for the reference's propagation entry points.

Same names, argument order and return conventions as the reference module
``utils`` (utils.py:54-96), so that ``model/SGNN.py``'s call sites
(``SGNN.py:234,301,404,489``) work unchanged when they import these instead:

    adjacency_tensor = csr_to_sparse_Tensor(scipy_adjacency, device)     # once
    processed_X = process_data_with_adjacency_high_order(adjacency_tensor, X, device, order=k)

``csr_to_sparse_Tensor`` returns a :class:`~sgnn_amd.norm_adj.NormAdj` handle
instead of a torch sparse COO tensor; the reference's callers treat the
adjacency as opaque apart from ``.shape[0]`` (model/SGNN.py:33), which the handle
provides.  A real torch sparse COO tensor is accepted too and converted on
first use.
"""
from __future__ import annotations

import torch

from .norm_adj import NormAdj

_ATTR = "_sgnn_norm_adj"


def coo_to_sparse_Tensor(coo_mat, device):
    """scipy COO -> device-resident adjacency handle (utils.py:59-62)."""
    return NormAdj.from_scipy(coo_mat, device)


def csr_to_sparse_Tensor(csr_mat, device):
    """scipy CSR/CSC/COO -> device-resident adjacency handle (utils.py:54-56)."""
    return NormAdj.from_scipy(csr_mat, device)


def _handle(adjacency, device):
    if isinstance(adjacency, NormAdj):
        return adjacency
    if isinstance(adjacency, torch.Tensor) and adjacency.layout == torch.sparse_coo:
        h = getattr(adjacency, _ATTR, None)
        if h is None:
            h = NormAdj.from_torch_sparse(adjacency, device)
            try:
                setattr(adjacency, _ATTR, h)
            except Exception:            # pragma: no cover - tensors normally accept attributes
                pass
        return h
    raise TypeError("adjacency must be a NormAdj handle or a torch sparse COO tensor, "
                    f"got {type(adjacency).__name__}")


def process_data_with_adjacency_high_order(adjacency, X, device, order=1):
    """``(D^-1/2 (A+I) D^-1/2)^order X`` (utils.py:81-96).

    Returns a new contiguous fp32 tensor on ``device``; ``X`` is not modified
    (for ``order == 0`` the reference returns ``X`` itself, and so does this).
    """
    h = _handle(adjacency, device)
    if not isinstance(X, torch.Tensor):
        raise TypeError("X must be a torch.Tensor")
    if X.device != h.device:
        X = X.to(h.device)
    if X.dtype != torch.float32:
        X = X.to(torch.float32)
    if X.requires_grad:
        X = X.detach()                   # the reference never differentiates through this call
    return h.propagate(X, order)


def process_data_with_adjacency(adjacency, X, device):
    """order=1 wrapper (utils.py:77-78)."""
    return process_data_with_adjacency_high_order(adjacency, X, device, order=1)
