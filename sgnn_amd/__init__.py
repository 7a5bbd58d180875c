"""sgnn_amd — MI355X-native sparse propagation for SGNN / SGC / GCN.

Public surface (mirrors the reference's entry points for this path):
  utils.csr_to_sparse_Tensor, utils.coo_to_sparse_Tensor,
  utils.process_data_with_adjacency, utils.process_data_with_adjacency_high_order,
  nn.SGConv, nn.GCNConv, and the NormAdj handle they share.
"""
from .norm_adj import NormAdj  # noqa: F401
from . import utils, nn  # noqa: F401

__all__ = ["NormAdj", "utils", "nn"]
