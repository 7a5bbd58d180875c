"""ctypes binding of libsgnn_hip.so (declared in include/sgnn_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc,
``--offload-arch=gfx950``).  There is no CPU fallback: if the shared object is
missing or a GPU entry point is called without a GPU the caller gets an
exception, never a silently slower path.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsgnn_hip.so")

# constants of include/sgnn_hip.h
STRIP_DIAG = 1
DROP_ZEROS = 2
SYMMETRIZE = 4
CLAMP_ONE = 8
SAMPLE_SKIP_SELF = 1
ADAM_STATE_WORDS = 1026
LOOPS_NONE, LOOPS_ADD, LOOPS_REMAINING = 0, 1, 2
EPI_SCALE_ROW, EPI_SCALE_ROW2 = 0, 1
# tile-group operand state (sgnn_tilegroup_state_bytes): 32-bit words; the per-feature maxima follow the header
TG_STATE_FLAG, TG_STATE_TINY, TG_STATE_TINY_LIMIT, TG_STATE_HEAD_WORDS = 1, 4, 6, 16
# words 2 and 3: the exact max / split passes ran behind a provisional split; the maxima are a sample's
TG_STATE_REDO, TG_STATE_SAMPLED = 2, 3

# name -> (restype, argtypes); every symbol the header declares
SIGNATURES = {
    "sgnn_version": (c_int, []),
    "sgnn_tune_set": (c_int, [c_char_p, c_int]),
    "sgnn_last_error": (c_char_p, []),
    "sgnn_kernel_time_ms": (c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int64)]),
    "sgnn_build_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sgnn_build_norm_csr": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    "sgnn_transpose_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sgnn_csr_transpose": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_size_t, c_void_p]),
    "sgnn_spmm_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "sgnn_spmm_csr_f32": (c_int, [c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_size_t,
                                  c_void_p]),
    "sgnn_spmm_csr_noalloc_f32": (c_int, [c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "sgnn_spmm_chain_f32": (c_int, [c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sgnn_rowgroup_config": (c_int, [c_int64]),
    "sgnn_rowgroup_config_for": (c_int, [c_int64, c_int64, c_int64]),
    "sgnn_rowgroup_plan_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "sgnn_rowgroup_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "sgnn_rowgroup_build": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_size_t,
                                    c_void_p, c_size_t, c_void_p]),
    "sgnn_spmm_rowgroup_workspace_bytes": (c_size_t, [c_int64, c_int, c_int64]),
    "sgnn_rowgroup_weights_bytes": (c_size_t, [c_int64, c_int]),
    "sgnn_rowgroup_weights": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t,
                                      c_int64, c_void_p, c_size_t, c_void_p]),
    "sgnn_spmm_rowgroup_f32": (c_int, [c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_size_t, c_void_p,
                                       c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                       ctypes.c_float, c_void_p, c_int64, c_void_p, c_int64,
                                       c_void_p, c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "sgnn_tilegroup_debug_counters": (c_int, [c_void_p]),
    "sgnn_tilegroup_debug_spans": (c_int, [c_void_p, c_int64]),
    "sgnn_tilegroup_build_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sgnn_tilegroup_plan_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int64, c_int64]),
    "sgnn_tilegroup_analyze": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sgnn_tilegroup_fill": (c_int, [c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p, c_size_t,
                                    c_void_p, c_size_t, c_void_p]),
    "sgnn_tilegroup_planes_bytes": (c_size_t, [c_int64, c_int64]),
    "sgnn_tilegroup_state_bytes": (c_size_t, [c_int64]),
    "sgnn_tilegroup_amax_f32": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "sgnn_tilegroup_planes_f32": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                          c_void_p, c_void_p]),
    "sgnn_tilegroup_split_f32": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p,
                                         c_void_p]),
    "sgnn_spmm_tilegroup_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sgnn_spmm_tilegroup_f32": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_size_t, c_int64, c_int64, c_int64,
                                        c_int64, c_int64, c_void_p, c_float, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p,
                                        c_int64, c_int, c_void_p, c_size_t, c_int64, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int64, c_void_p, c_size_t,
                                        c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int,
                                        c_void_p]),
    "sgnn_spmm_tilegroup_slices_f32": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_size_t, c_int64, c_int64, c_int64,
                                               c_int64, c_int64, c_void_p, c_float, c_void_p, c_void_p, c_size_t, c_void_p,
                                               c_void_p, c_int64, c_int, c_void_p, c_size_t, c_int64, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_size_t,
                                               c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int,
                                               c_int64, c_int64, c_int, c_void_p]),
    "sgnn_tilegroup_chain_f32": (c_int, [c_int, c_int64, c_int64, c_void_p, c_size_t, c_int64, c_int64, c_int64, c_int64,
                                         c_int64, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                         c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t,
                                         c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sgnn_row_scale_f32": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "sgnn_stream_copy_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "sgnn_add_scale_f32": (c_int, [c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p,
                                   c_int64, c_void_p]),
    "sgnn_gae_targets_f32": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                     c_void_p, ctypes.c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "sgnn_gae_loss_grad_f32": (c_int, [c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "sgnn_batch_indices": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sgnn_adam_step_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                   c_float, c_float, c_float, c_void_p, c_void_p]),
    "sgnn_loss_workspace_bytes": (c_size_t, [c_int64]),
    "sgnn_head_xent_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "sgnn_head_xent_f32": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                   c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_size_t,
                                   c_void_p]),
    "sgnn_sum_f32": (c_int, [c_int64, c_void_p, c_float, c_void_p, c_void_p]),
    "sgnn_batch_project_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int]),
    "sgnn_batch_project_input_grad_f32": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                                  c_void_p, c_int64, c_float, c_void_p, c_int64, c_void_p, c_size_t,
                                                  c_void_p]),
    "sgnn_batch_project_f32": (c_int, [c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_float, c_void_p, c_int64,
                                       c_int64, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "sgnn_batch_project_grad_f32": (c_int, [c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_float, c_void_p,
                                            c_int64, c_int64, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "sgnn_softmax_xent_f32": (c_int, [c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                      c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "sgnn_sq_err_f32": (c_int, [c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_float,
                                c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "sgnn_sample_block_f32": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int,
                                      c_void_p, c_int64, c_void_p]),
}

_lib = None


class SgnnHipError(RuntimeError):
    pass


def load():
    """Load the shared library once and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SgnnHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). sgnn_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().sgnn_last_error().decode("utf-8", "replace")
        raise SgnnHipError(f"libsgnn_hip error {rc}: {msg}")


def kernel_time_ms():
    """(sum of the main SpMM kernels' durations in ms, launches) since the last read; needs
    ``sgnn_tune_set(b"time_kernel", 1)``."""
    total, launches = ctypes.c_double(0), c_int64(0)
    check(load().sgnn_kernel_time_ms(ctypes.byref(total), ctypes.byref(launches)))
    return total.value, launches.value


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def current_stream(device):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)
