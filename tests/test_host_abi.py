"""CPU-side checks of the boundary: the shared library loads without a GPU and
exports exactly the entry points include/sgnn_hip.h declares; the host layer
refuses to run without a GPU instead of falling back to anything."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "sgnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sgnn_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from sgnn_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert names, "no declarations found in include/sgnn_hip.h"
    assert sorted(_lib.SIGNATURES) == names          # the ctypes table and the header agree
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} is declared but not exported"
    assert lib.sgnn_version() >= 100
    assert lib.sgnn_last_error() == b""


def test_workspace_queries_need_no_gpu():
    from sgnn_amd import _lib
    lib = _lib.load()
    small = lib.sgnn_spmm_workspace_bytes(1000, 5000, 64)
    big = lib.sgnn_spmm_workspace_bytes(232_965, 114_848_857, 602)
    assert 0 < small < big < (1 << 31)
    assert lib.sgnn_build_workspace_bytes(1000, 5000) > 5000 * 24
    assert lib.sgnn_transpose_workspace_bytes(1000, 5000) > 0


def test_argument_errors_are_reported_not_crashes():
    from sgnn_amd import _lib
    lib = _lib.load()
    # invalid arguments are rejected before any HIP call, so this is safe without a GPU
    rc = lib.sgnn_spmm_csr_f32(4, 4, 4, -1, None, None, None, None, None, None, 4, None, 4, 0, None, 0, None)
    assert rc == -1 and b"negative" in lib.sgnn_last_error()
    rc = lib.sgnn_spmm_csr_f32(4, 4, 4, 8, None, None, None, None, None, None, 8, None, 8, 0, None, 0, None)
    assert rc == -1 and b"null" in lib.sgnn_last_error()
    rc = lib.sgnn_spmm_chain_f32(0, 4, 4, 8, None, None, None, None, None, 8, None, 8, None, None, 0, None)
    assert rc == -1
    rc = lib.sgnn_build_norm_csr(0, 0, None, None, None, 0, 1, 0, None, None, None, None, None, None, None, 0, None)
    assert rc == -1
    with pytest.raises(_lib.SgnnHipError):
        _lib.check(rc)
    rc = lib.sgnn_build_norm_csr(4, 2, None, None, None, 64, 1, 0, None, None, None, None, None, None, None, 0, None)
    assert rc == -1 and b"flag" in lib.sgnn_last_error()
    rc = lib.sgnn_sample_block_f32(4, 4, None, None, None, 3, None, 0, None, 3, None)
    assert rc == -1 and b"null" in lib.sgnn_last_error()
    rc = lib.sgnn_sample_block_f32(4, 4, None, None, None, 3, None, 2, None, 3, None)
    assert rc == -1 and b"flag" in lib.sgnn_last_error()
    rc = lib.sgnn_sample_block_f32(4, 4, None, None, None, 3, None, 0, None, 2, None)
    assert rc == -1 and b"ldo" in lib.sgnn_last_error()
    assert lib.sgnn_sample_block_f32(4, 4, None, None, None, 0, None, 0, None, 0, None) == 0
    # batch indices / auto-encoder loss / group weights: argument checks come before any launch
    assert lib.sgnn_batch_indices(5, 6, None, None, None, None) == -1 and b"b <= n" in lib.sgnn_last_error()
    assert lib.sgnn_batch_indices(5, 2, None, None, None, None) == -1 and b"null" in lib.sgnn_last_error()
    assert lib.sgnn_gae_targets_f32(4, 4, None, None, None, None, None, 3, None, 1.5, None,
                                    None, None, None, None, None) == -1 and b"mask_rate" in lib.sgnn_last_error()
    assert lib.sgnn_gae_targets_f32(4, 4, None, None, None, None, None, 0, None, 0.2, None,
                                    None, None, None, None, None) == 0
    assert lib.sgnn_gae_loss_grad_f32(3, None, 2, None, None, None, None, None, None, None, None) == -1
    assert lib.sgnn_rowgroup_weights_bytes(1000, 16) >= 1000 * 16 * 4
    assert lib.sgnn_rowgroup_weights_bytes(1000, 5) == 0
    assert lib.sgnn_rowgroup_weights(4, 4, None, None, None, 16, None, 0, 4, None, 0, None) == -1
    assert lib.sgnn_rowgroup_config(16) == 0 and lib.sgnn_rowgroup_config(64) == 8
    assert lib.sgnn_rowgroup_config(602) == 16 and lib.sgnn_rowgroup_config(3703) == 16
    # few rows: not enough groups of 16 to occupy the chip
    assert lib.sgnn_rowgroup_config_for(19717, 108395, 500) == 8
    assert lib.sgnn_rowgroup_config_for(232965, 114848857, 602) == 16
    assert lib.sgnn_rowgroup_config_for(29121, 14356107, 602) == 16          # few rows, many entries
    assert lib.sgnn_rowgroup_config_for(19717, 108395, 16) == 0 and lib.sgnn_rowgroup_config_for(232965, 10 ** 8, 64) == 8


def test_no_cpu_fallback():
    from sgnn_amd import NormAdj, utils
    from sgnn_amd._lib import SgnnHipError
    with pytest.raises(SgnnHipError):
        NormAdj.from_coo(np.array([0, 1]), np.array([1, 0]), None, 2, "cpu")
    if not torch.cuda.is_available():
        with pytest.raises(SgnnHipError):
            NormAdj.from_coo(np.array([0, 1]), np.array([1, 0]), None, 2, "cuda:0")
    with pytest.raises(TypeError):
        utils.process_data_with_adjacency_high_order(object(), torch.zeros(2, 2), "cpu")


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sgnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text, f"{f} mentions the oracle"
