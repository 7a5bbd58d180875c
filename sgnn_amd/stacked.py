"""Decoupled stacked-GNN training on top of the propagation boundary.

This is the caller side of the hot path: the stacked models of the reference —
``SupervisedStackedGNN`` with embedding-GCN layers (the only layer type its
classification runner builds, GNN_tasks.py:76-87) and ``StackedGNN`` with graph
auto-encoder layers (the clustering runner, GNN_tasks.py:325-356) — restated in
plain PyTorch so that the end-to-end accuracy / NMI of the reference can be
reproduced above ``sgnn_amd.utils``.  The numerics that matter for parity follow
the reference:

* a layer is ``act(inner_act(X' U) W)`` with ``X' = A_hat^order X``
  (model/SGNN.py:73-78, 404-405), ``U`` starting at the identity;
* forward sweeps train ``W`` (and the label head ``Wt``) with ``U`` frozen,
  backward sweeps also train ``U`` of every layer but the first
  (model/SGNN.py:567-600), targets being the next layer's expected input
  ``X U`` (model/SGNN.py:442, 576-578, 599);
* losses: cross-entropy through ``Wt`` without a target, ``eta * ||E - T||_F^2 / m``
  with one (model/SGNN.py:80-86, 388-399), a second target term on a batch drawn
  from all nodes (model/SGNN.py:425-429), ridge penalty ``lam * (|W|^2+|b|^2+|U|^2)``
  (model/SGNN.py:88-92), a fresh Adam per ``run`` (model/SGNN.py:417), batches drawn
  as the head of a random permutation (model/SGNN.py:115-117);
* auto-encoder layers reconstruct the dense sub-adjacency of the batch through
  ``sigmoid(clamp(Z Z^T, +-40))`` under the positive-reweighted cross-entropy with a
  random symmetric ignore mask merged with the overlook matrix
  (model/SGNN.py:138-164, 209-226, 682-702) and a lasso penalty
  (model/SGNN.py:93-96, 543); the batch's sub-adjacency is read from the CSR on
  the GPU (``NormAdj.sample_block``) instead of scipy fancy indexing plus a
  host-to-device copy per iteration (model/SGNN.py:167-175).

What differs, on purpose (these are the residency and launch-overhead items the
reference leaves on the table): the Adam iterations of a layer run are captured once into
a HIP graph and replayed (``_Layer._optimise``); features, targets and embeddings stay on
the GPU instead of bouncing through host memory on every call (model/SGNN.py:404,
442-443), and ``A_hat^k X`` of a layer is computed once per distinct input instead of on
every ``run`` — the first layer's input never changes, so 2*BP_count of its 2*BP_count+1
propagations disappear.

The label head ``Wt`` is a trained parameter on every device.  In the reference it is
trained only on the CPU: with a GPU device, ``self.Wt = get_weight_initial(...).to(device)``
(model/SGNN.py:376) makes a copy that is not a leaf and not in ``parameters()``, so Adam
never updates it there and the head stays a random projection.  The end-to-end golden was
captured from the reference on the CPU, where it trains; accuracies of a reference run on
CUDA are not comparable with the ones produced here.

An optimiser iteration itself runs on the library's own kernels where its shape allows:
the batch projection with the row gather and the inner activation folded in
(``_BatchProject``, ``_DenseProject``), the label head with its loss and gradients
(``_HeadXent``), the target error (``_SqErr``), the reconstruction loss
(``_FusedReconstruction``) and Adam with the ridge / LASSO penalty applied through the
gradient (``_FusedAdam``); each has the stock torch operators as the route for shapes or
dtypes it does not cover, and ``tests/test_gpu_train_step.py`` holds them to those
operators.  The penalties' values are therefore not part of any loss value returned here.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as Fn

from . import _lib
from . import utils as prop


class Func(torch.nn.Module):
    """Activation wrapper with the reference's signature (model/SGNN.py:706-713)."""

    def __init__(self, func, **params):
        super().__init__()
        self.func, self.params = func, params

    def forward(self, X):
        return X if self.func is None else self.func(X, **self.params)


def get_activation(spec):
    """Activation strings of the reference's configs (GNN_tasks.py:387-404)."""
    if "tanh" in spec:
        return Func(torch.tanh)
    if "sigmoid" in spec:
        return Func(torch.sigmoid)
    if "linear" in spec:
        return Func(None)
    if "leaky" in spec:
        return Func(Fn.leaky_relu, negative_slope=float(spec.split("=")[1]))
    if spec == "relu":
        return Func(Fn.relu)
    raise ValueError(f"unknown activation {spec!r}")


class LayerParam:
    """Per-layer hyper-parameters (model/SGNN.py:729-758)."""
    GAE, GCN, EGCN = 0, 1, 2
    MASK_RATE = "mask_rate"

    def __init__(self, neurons, inner_act=None, act=None, gnn_type=EGCN, **kwargs):
        self.neurons = neurons
        self.inner_activation = inner_act if inner_act is not None else Func(None)
        self.activation = act if act is not None else Func(None)
        self.gnn_type = gnn_type
        self.extra_params = kwargs

    def get(self, key, default):
        return self.extra_params.get(key, default)


def _glorot(rows, cols, device):
    bound = math.sqrt(6.0 / (rows + cols))                       # utils.py:65-68
    return torch.nn.Parameter((torch.rand(rows, cols, device=device) * 2 - 1) * bound)


class _FusedAdam:
    """``torch.optim.Adam`` (default betas / eps, coupled weight decay per group) for a
    handful of small tensors as ONE launch per step (``sgnn_adam_step_f32``) instead of a
    multi-tensor pass per parameter group plus step-counter kernels; a group's ``l1_decay``
    adds ``l1_decay * sign(p)`` to the gradient (the LASSO penalty).  The step number lives
    in device memory and is advanced by the kernel, so a captured iteration keeps counting
    when the HIP graph is replayed.  Parameters whose ``.grad`` is None are skipped, as
    torch does."""
    MAX_TENSORS = 16

    def __init__(self, groups, lr, device, betas=(0.9, 0.999), eps=1e-8):
        self.entries = [(q, float(g.get("weight_decay", 0.0)), float(g.get("l1_decay", 0.0)))
                        for g in groups for q in g["params"]]
        for q, _, _ in self.entries:
            if q.dtype != torch.float32 or not q.is_contiguous():
                raise ValueError("fused Adam needs contiguous fp32 parameters")
        self.lr, self.betas, self.eps, self.device = float(lr), betas, float(eps), device
        self.exp_avg = [torch.zeros_like(e[0]) for e in self.entries]
        self.exp_avg_sq = [torch.zeros_like(e[0]) for e in self.entries]
        self.state = torch.zeros(_lib.ADAM_STATE_WORDS, dtype=torch.int64, device=device)

    def zero_grad(self, set_to_none=True):
        for q, _, _ in self.entries:
            if set_to_none:
                q.grad = None
            elif q.grad is not None:
                q.grad.zero_()

    @torch.no_grad()
    def step(self):
        import ctypes
        live = [k for k, e in enumerate(self.entries) if e[0].grad is not None]
        if not live:
            return
        for k in live:
            g = self.entries[k][0].grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                raise ValueError("fused Adam needs contiguous fp32 gradients")
        n = len(live)
        ptrs = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])          # noqa: E731
        sizes = (ctypes.c_int64 * n)(*[self.entries[k][0].numel() for k in live])
        decay = (ctypes.c_float * n)(*[self.entries[k][1] for k in live])
        l1 = (ctypes.c_float * n)(*[self.entries[k][2] for k in live])
        _lib.check(_lib.load().sgnn_adam_step_f32(
            n, ptrs([self.entries[k][0] for k in live]), ptrs([self.entries[k][0].grad for k in live]),
            ptrs([self.exp_avg[k] for k in live]), ptrs([self.exp_avg_sq[k] for k in live]), sizes, decay, l1,
            self.lr, self.betas[0], self.betas[1], self.eps, _lib.ptr(self.state), _lib.current_stream(self.device)))


class _BatchProject(torch.autograd.Function):
    """``act(X[idx]) @ W`` with the gather and the inner activation folded into the operand
    load (``sgnn_batch_project_f32``); the backward forms ``dW`` the same way.  ``X`` is data
    (no gradient)."""

    @staticmethod
    def forward(ctx, X, idx, W, slope):
        b, (F, N) = idx.numel(), W.shape
        E = torch.empty(b, N, dtype=torch.float32, device=X.device)
        lib = _lib.load()
        ws = torch.empty(lib.sgnn_batch_project_workspace_bytes(b, F, N, 0) // 4, dtype=torch.float32, device=X.device)
        _lib.check(lib.sgnn_batch_project_f32(X.shape[0], F, _lib.ptr(X), X.stride(0), b, _lib.ptr(idx),
                                              float(slope), _lib.ptr(W), W.stride(0), N, _lib.ptr(E), N,
                                              _lib.ptr(ws), ws.numel() * 4, _lib.current_stream(X.device)))
        ctx.save_for_backward(X, idx)
        ctx.slope, ctx.shape = float(slope), (F, N)
        return E

    @staticmethod
    def backward(ctx, dE):
        X, idx = ctx.saved_tensors
        F, N = ctx.shape
        dE = dE.contiguous()
        dW = torch.empty(F, N, dtype=torch.float32, device=X.device)
        lib, b = _lib.load(), idx.numel()
        ws = torch.empty(lib.sgnn_batch_project_workspace_bytes(b, F, N, 1) // 4, dtype=torch.float32, device=X.device)
        _lib.check(lib.sgnn_batch_project_grad_f32(X.shape[0], F, _lib.ptr(X), X.stride(0), b, _lib.ptr(idx),
                                                   ctx.slope, _lib.ptr(dE), N, N, _lib.ptr(dW), N, _lib.ptr(ws),
                                                   ws.numel() * 4, _lib.current_stream(X.device)))
        return None, None, dW, None


class _DenseProject(torch.autograd.Function):
    """``act(A) @ W`` for a dense batch ``A`` that needs its own gradient (``A = X[idx] @ U``
    while U is trained): the projection kernel with no index list, ``dW`` the same way and
    ``dA = (dE @ W.T) * act'(A)`` from its transposed-operand variant."""

    @staticmethod
    def forward(ctx, A, W, slope):
        A = A.contiguous()
        b, (F, N) = A.shape[0], W.shape
        lib, dev = _lib.load(), A.device
        E = torch.empty(b, N, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.sgnn_batch_project_workspace_bytes(b, F, N, 0) // 4, dtype=torch.float32, device=dev)
        _lib.check(lib.sgnn_batch_project_f32(b, F, _lib.ptr(A), F, b, None, float(slope), _lib.ptr(W), W.stride(0), N,
                                              _lib.ptr(E), N, _lib.ptr(ws), ws.numel() * 4, _lib.current_stream(dev)))
        ctx.save_for_backward(A, W)
        ctx.slope = float(slope)
        return E

    @staticmethod
    def backward(ctx, dE):
        A, W = ctx.saved_tensors
        b, (F, N) = A.shape[0], W.shape
        lib, dev = _lib.load(), A.device
        dE = dE.contiguous()
        st = _lib.current_stream(dev)
        dA = dW = None
        if ctx.needs_input_grad[1]:
            dW = torch.empty(F, N, dtype=torch.float32, device=dev)
            ws = torch.empty(lib.sgnn_batch_project_workspace_bytes(b, F, N, 1) // 4, dtype=torch.float32, device=dev)
            _lib.check(lib.sgnn_batch_project_grad_f32(b, F, _lib.ptr(A), F, b, None, ctx.slope, _lib.ptr(dE), N, N,
                                                       _lib.ptr(dW), N, _lib.ptr(ws), ws.numel() * 4, st))
        if ctx.needs_input_grad[0]:
            dA = torch.empty(b, F, dtype=torch.float32, device=dev)
            ws = torch.empty(lib.sgnn_batch_project_workspace_bytes(b, F, N, 2) // 4, dtype=torch.float32, device=dev)
            _lib.check(lib.sgnn_batch_project_input_grad_f32(b, F, N, _lib.ptr(dE), N, _lib.ptr(W), W.stride(0),
                                                             _lib.ptr(A), F, ctx.slope, _lib.ptr(dA), F, _lib.ptr(ws),
                                                             ws.numel() * 4, st))
        return dA, dW, None


def _activation_slope(act):
    """The slope s of ``act(x) = x if x > 0 else s * x`` for the activations that have that
    form (linear, relu, leaky_relu), else None."""
    if not isinstance(act, Func):
        return None
    if act.func is None:
        return 1.0
    if act.func is Fn.relu and not act.params:
        return 0.0
    if act.func is Fn.leaky_relu and set(act.params) <= {"negative_slope"}:
        return float(act.params.get("negative_slope", 0.01))
    return None


class _SoftmaxXent(torch.autograd.Function):
    """Mean cross-entropy of ``logits`` against ``labels[idx]`` and its gradient in one
    launch (``sgnn_softmax_xent_f32``)."""

    @staticmethod
    def forward(ctx, logits, labels, idx, unit):
        logits = logits.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        dz = torch.empty_like(logits)
        B, C = logits.shape
        ws = torch.empty(B, dtype=torch.float32, device=logits.device)
        _lib.check(_lib.load().sgnn_softmax_xent_f32(B, C, _lib.ptr(logits), C, _lib.ptr(labels), labels.numel(),
                                                     _lib.ptr(idx) if idx is not None else None, _lib.ptr(loss),
                                                     _lib.ptr(dz), C, _lib.ptr(ws), ws.numel() * 4,
                                                     _lib.current_stream(logits.device)))
        ctx.save_for_backward(dz)
        ctx.unit = unit
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        (dz,) = ctx.saved_tensors
        return (dz if ctx.unit else dz * grad_loss), None, None, None


class _HeadXent(torch.autograd.Function):
    """``cross_entropy(E @ Wt, labels[idx])`` with both gradients from ``sgnn_head_xent_f32``
    (two launches instead of GEMM, softmax, nll and two backward GEMMs)."""

    @staticmethod
    def forward(ctx, E, Wt, labels, idx, unit):
        E = E.contiguous()
        B, D = E.shape
        C = Wt.shape[1]
        dev = E.device
        lib = _lib.load()
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dE = torch.empty_like(E)
        dWt = torch.empty(D, C, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.sgnn_head_xent_workspace_bytes(B, D, C) // 4, dtype=torch.float32, device=dev)
        _lib.check(lib.sgnn_head_xent_f32(B, D, C, _lib.ptr(E), D, _lib.ptr(Wt), Wt.stride(0), _lib.ptr(labels),
                                          labels.numel(), _lib.ptr(idx) if idx is not None else None, _lib.ptr(loss),
                                          _lib.ptr(dE), D, _lib.ptr(dWt), C, _lib.ptr(ws), ws.numel() * 4,
                                          _lib.current_stream(dev)))
        ctx.save_for_backward(dE, dWt)
        ctx.unit = unit
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        dE, dWt = ctx.saved_tensors
        if ctx.unit:
            return dE, dWt, None, None, None
        return dE * grad_loss, dWt * grad_loss, None, None, None


def _head_cross_entropy(E, Wt, labels, idx, unit=False):
    """``cross_entropy(E @ Wt, labels[idx])``."""
    if (E.is_cuda and E.dtype == torch.float32 and E.dim() == 2 and Wt.dtype == torch.float32 and Wt.stride(1) == 1
            and labels.dtype == torch.int64 and labels.is_contiguous() and E.shape[0] > 0
            and E.shape[1] <= 256 and Wt.shape[1] <= 64
            and 4 * (E.shape[1] * Wt.shape[1] + 8 * E.shape[1] + 8 * Wt.shape[1]) <= 65536):
        return _HeadXent.apply(E, Wt, labels, idx, unit)
    return _cross_entropy(E.matmul(Wt), labels, idx, unit)


class _SqErr(torch.autograd.Function):
    """``scale * ||a - target[idx]||_F^2`` and its gradient in one launch (``sgnn_sq_err_f32``)."""

    @staticmethod
    def forward(ctx, a, target, idx, scale, unit):
        a = a.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        grad = torch.empty_like(a)
        B, D = a.shape
        ws = torch.empty(B, dtype=torch.float32, device=a.device)
        _lib.check(_lib.load().sgnn_sq_err_f32(B, D, _lib.ptr(a), D, _lib.ptr(target), target.stride(0),
                                               target.shape[0], _lib.ptr(idx) if idx is not None else None,
                                               float(scale), _lib.ptr(loss), _lib.ptr(grad), D, _lib.ptr(ws),
                                               ws.numel() * 4, _lib.current_stream(a.device)))
        ctx.save_for_backward(grad)
        ctx.unit = unit
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        (grad,) = ctx.saved_tensors
        return (grad if ctx.unit else grad * grad_loss), None, None, None, None


_FUSED_LOSS_MAX = 1 << 30      # elements


# ``unit=True`` on the fused losses is a promise by the caller: the returned loss enters the
# objective with coefficient 1 and ``backward()`` starts at the objective (gradient 1), as
# ``_Layer._optimise`` does with the plain sums the layers build.  The backward then hands
# the gradient it already computed on, instead of launching a multiplication by 1.
def _cross_entropy(logits, labels, idx, unit=False):
    """``cross_entropy(logits, labels[idx])`` (mean over rows)."""
    if (logits.is_cuda and logits.dtype == torch.float32 and labels.dtype == torch.int64 and labels.is_contiguous()
            and 0 < logits.numel() <= _FUSED_LOSS_MAX):
        return _SoftmaxXent.apply(logits, labels, idx, unit)
    return Fn.cross_entropy(logits, labels[idx])


def _squared_error(a, target, idx, scale, unit=False):
    """``scale * ((a - target[idx]) ** 2).sum()``."""
    if (a.is_cuda and a.dtype == torch.float32 and target.dtype == torch.float32 and target.dim() == 2
            and target.stride(1) == 1 and 0 < a.numel() <= _FUSED_LOSS_MAX):
        return _SqErr.apply(a, target, idx, scale, unit)
    return scale * (a - target[idx]).pow(2).sum()


class _Layer(torch.nn.Module):
    """What every decoupled layer shares (SingleLayerGNN, model/SGNN.py:18-117)."""

    def __init__(self, adjacency, input_dim, embedding_dim, p, device):
        super().__init__()
        self.adjacency, self.device = adjacency, device
        self.order = p.get("order", 1)
        self.lam = p.get("lam", 0)
        self.lr = p.get("learning_rate", 0.01)
        self.max_iter = p.get("max_iter", 10)
        self.batch_size = p.get("batch_size", 64)
        self.inner_activation, self.activation = p.inner_activation, p.activation
        self.W = _glorot(input_dim, embedding_dim, device)
        self.b = _glorot(1, embedding_dim, device)
        self.U = torch.nn.Parameter(torch.eye(input_dim, device=device), requires_grad=False)
        self.expected_X = None
        self._u_identity = True          # U still holds the identity it was created with / reset to
        self.use_graph = p.get("use_graph", True)
        self._rng_state = None
        self._cache_in, self._cache_version, self._cache_val = None, -1, None

    def set_training_direction(self, is_backward, reset_backward=True):
        self.U.requires_grad = is_backward
        if is_backward:
            self._u_identity = False     # about to be trained
        elif reset_backward:
            with torch.no_grad():
                self.U.copy_(torch.eye(self.U.shape[0], device=self.device))
            self._u_identity = True

    def forward(self, X):
        # X . I = X exactly in fp32 (x * 1 plus exact zeros), so the product is skipped
        # while U is the untrained identity: forward sweeps and plain stacked training
        XU = X if self._u_identity else X.matmul(self.U)
        return self.activation(self.inner_activation(XU).matmul(self.W))

    def embed_batch(self, X, idx):
        """``self(X[idx])``; one launch for gather, inner activation and projection while U
        is the identity and the inner activation is linear / relu / leaky_relu."""
        slope = _activation_slope(self.inner_activation)
        if (self._u_identity and slope is not None and X.is_cuda and X.dtype == torch.float32 and X.dim() == 2
                and X.stride(1) == 1 and not X.requires_grad and self.W.is_contiguous() and idx.numel() > 0):
            return self.activation(_BatchProject.apply(X, idx, self.W, slope))
        if (not self._u_identity and X.is_cuda and X.dtype == torch.float32 and X.dim() == 2 and X.stride(1) == 1
                and not X.requires_grad and self.U.is_contiguous() and idx.numel() > 0):
            XU = _BatchProject.apply(X, idx, self.U, 1.0)          # gather folded into X[idx] . U and into dU
            if slope is not None and self.W.is_contiguous():
                return self.activation(_DenseProject.apply(XU, self.W, slope))
            return self.activation(self.inner_activation(XU).matmul(self.W))
        return self(X[idx])

    def propagated(self, X):
        """``A_hat^order X``, recomputed only when the input changes."""
        # keyed on the tensor object (kept alive here, so its storage cannot be reused
        # by another tensor) and its in-place version counter
        if X is not self._cache_in or X._version != self._cache_version:
            self._cache_val = prop.process_data_with_adjacency_high_order(self.adjacency, X, self.device,
                                                                          order=self.order)
            self._cache_in, self._cache_version = X, X._version
        return self._cache_val

    def _lasso(self):
        return self.W.abs().sum() + self.b.abs().sum() + self.U.abs().sum()

    def _batch(self, rows, row_map=None, out=None):
        """Indices of a batch drawn without replacement: the head of a pseudo-random
        permutation of ``range(rows)`` (model/SGNN.py:115-117 takes the head of
        ``torch.randperm``), from ``sgnn_batch_indices`` — one small kernel instead of a
        sort of ``rows`` keys, and it draws afresh on every replay of a HIP graph.
        ``row_map`` (int64[rows]) turns the drawn positions into the ids they stand for;
        ``out`` receives the indices (a slice of a larger index vector, say)."""
        if self._rng_state is None:      # seeded from torch's generator, so torch.manual_seed governs it
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            self._rng_state = torch.tensor([seed, 0, 0], dtype=torch.int64, device=self.device)
        b = min(self.batch_size, rows)
        if out is None:
            out = torch.empty(b, dtype=torch.int64, device=self.device)
        assert out.numel() == b and out.is_contiguous() and (row_map is None or row_map.numel() == rows)
        _lib.check(_lib.load().sgnn_batch_indices(rows, b, _lib.ptr(self._rng_state), _lib.ptr(row_map),
                                                  _lib.ptr(out), _lib.current_stream(self.device)))
        return out

    # Optimiser iterations recorded into one HIP graph and replayed.  An iteration is
    # ~100 kernels of a few microseconds each; issued one by one from Python it takes
    # 0.7-1.2 ms, nearly all of it launch overhead on the host.
    GRAPH_WARMUP = 3

    def _optimise(self, loss_fn, ridge=0.0, lasso=0.0):
        """``max_iter`` Adam steps on ``loss_fn()`` with a fresh optimiser (model/SGNN.py:417).

        ``ridge`` > 0 adds the penalty ``ridge * (|W|^2 + |b|^2 + |U|^2)`` of
        model/SGNN.py:88-92 through its gradient ``2 * ridge * p``, which is what Adam's
        (coupled) ``weight_decay`` adds to the gradient: the same update without the six
        reduction kernels and their backward per iteration.  ``lasso`` > 0 adds
        ``lasso * (|W|_1 + |b|_1 + |U|_1)`` (model/SGNN.py:93-96) the same way, as
        ``lasso * sign(p)`` on the gradient inside the fused update."""
        use_graph = self.use_graph and self.device.type == "cuda" and self.max_iter > 2 * self.GRAPH_WARMUP
        penalised = [self.W, self.b, self.U]
        others = [q for q in self.parameters() if all(q is not t for t in penalised)]
        groups = [dict(params=penalised, weight_decay=2.0 * ridge)]
        if others:
            groups.append(dict(params=others, weight_decay=0.0))
        lasso_in_loss = 0.0
        if self.device.type == "cuda" and sum(len(g["params"]) for g in groups) <= _FusedAdam.MAX_TENSORS:
            groups[0]["l1_decay"] = lasso
            opt = _FusedAdam(groups, self.lr, self.device)
        else:
            lasso_in_loss = lasso
            opt = torch.optim.Adam(groups, lr=self.lr, capturable=use_graph,
                                   fused=True if self.device.type == "cuda" else None)

        def step():
            loss = loss_fn()
            if lasso_in_loss:
                loss = loss + lasso_in_loss * self._lasso()
            loss.backward()
            opt.step()

        if not use_graph:
            for _ in range(self.max_iter):
                opt.zero_grad()
                step()
            return
        # the first steps run eagerly on a side stream (they create the optimiser state
        # and let the allocator settle), the rest replay one captured step
        cur = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(self.GRAPH_WARMUP):
                opt.zero_grad(set_to_none=True)
                step()
        cur.wait_stream(side)
        opt.zero_grad(set_to_none=True)
        done = self.GRAPH_WARMUP
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()                           # recorded, not run
        except RuntimeError as exc:
            # an activation or loss the caller plugged in may not be capturable; the
            # iterations are then issued one by one (same arithmetic, more launch overhead)
            import warnings
            warnings.warn(f"HIP graph capture of the optimiser iteration failed ({exc}); running eagerly")
            self.use_graph = False
            graph = None
        for _ in range(self.max_iter - done):
            if graph is not None:
                graph.replay()
            else:
                opt.zero_grad(set_to_none=True)
                step()

    def _finish(self, X, PX):
        with torch.no_grad():
            embedding = self(PX)
            self.expected_X = X if self._u_identity else X.matmul(self.U)
        return embedding


def _mask_to_rows(mask, device, n=None):
    """Node ids a training mask selects.  The reference fancy-indexes with the mask
    (model/SGNN.py:412-415), which takes a boolean mask (Planetoid loaders) as well as an
    index array (data_loader.py:133 passes ``train_index`` for the OGB graphs).  With ``n``
    (the number of nodes) the selection is checked as that indexing checks it: a boolean mask
    must have ``n`` entries, negative ids count from the end, ids outside ``[-n, n)`` raise
    ``IndexError``; an empty selection raises ``ValueError`` (the reference would fail later, in
    ``randperm(0)[:batch]`` feeding a mean over no rows).  The ids go to the kernels as raw row
    numbers, so nothing unchecked may pass here."""
    mask = torch.as_tensor(mask).to(device)
    if mask.dtype in (torch.bool, torch.uint8):          # torch indexes with a byte tensor as with a mask
        if mask.dim() != 1:
            raise ValueError("a boolean training mask must be 1-D")
        if n is not None and mask.numel() != n:
            raise IndexError(f"boolean training mask has {mask.numel()} entries for {n} nodes")
        rows = torch.nonzero(mask).squeeze(1)
    elif mask.dtype in (torch.int64, torch.int32, torch.int16, torch.int8):
        if mask.dim() != 1:
            raise ValueError("an index-array mask must be 1-D")
        rows = mask.long().contiguous()
        if n is not None and rows.numel() > 0:
            lo, hi = int(rows.min().item()), int(rows.max().item())
            if lo < -n or hi >= n:
                raise IndexError(f"training index {lo if lo < -n else hi} is out of bounds for {n} nodes")
            if lo < 0:
                rows = torch.where(rows < 0, rows + n, rows)
    else:
        raise TypeError(f"training mask must be boolean or an integer index array, got {mask.dtype}")
    if n is not None and rows.numel() == 0:
        raise ValueError("the training mask selects no node")
    return rows


class EmbeddingLayer(_Layer):
    """One decoupled layer with a label head (SingleLayerEmbeddingGCN, model/SGNN.py:357-467)."""

    def __init__(self, adjacency, labels, train_mask, input_dim, embedding_dim, n_class, p, device):
        super().__init__(adjacency, input_dim, embedding_dim, p, device)
        self.Wt = _glorot(embedding_dim, n_class, device)
        self.labels, self.train_mask = labels, train_mask

    def run(self, X, embedding_target=None, eta=1, train=True):
        PX = self.propagated(X)
        if train:
            # The reference samples rows of processed_X[mask] (model/SGNN.py:412-421); here the
            # drawn positions are mapped to node ids instead, so every batch reads PX, the labels
            # and the target directly and no masked copies are made.
            train_rows = _mask_to_rows(self.train_mask, self.device, PX.shape[0])
            n_train, n_all = train_rows.numel(), PX.shape[0]
            b1, b2 = min(self.batch_size, n_train), min(self.batch_size, n_all)

            def loss_fn():
                if embedding_target is None:
                    idx = self._batch(n_train, row_map=train_rows)
                    emb = self.embed_batch(PX, idx)
                    return _head_cross_entropy(emb, self.Wt, self.labels, idx, unit=True)
                # a batch of training nodes and a batch of all nodes against the target
                # (model/SGNN.py:421-428): drawn into one index vector, projected in one go
                both = torch.empty(b1 + b2, dtype=torch.int64, device=self.device)
                self._batch(n_train, row_map=train_rows, out=both[:b1])
                self._batch(n_all, out=both[b1:])
                emb = self.embed_batch(PX, both)
                if b1 == b2:
                    return _squared_error(emb, embedding_target, both, eta / b1, unit=True)
                return (_squared_error(emb[:b1], embedding_target, both[:b1], eta / b1, unit=True)
                        + _squared_error(emb[b1:], embedding_target, both[b1:], eta / b2, unit=True))

            self._optimise(loss_fn, ridge=self.lam)
        return self._finish(X, PX)

    @torch.no_grad()
    def predict(self, embedding):
        return embedding.matmul(self.Wt).argmax(dim=1)


class _FusedReconstruction(torch.autograd.Function):
    """Reconstruction loss of one batch through ``sgnn_gae_targets_f32`` /
    ``sgnn_gae_loss_grad_f32``: two kernels and a GEMM instead of ~60 elementwise and
    reduction launches over the b x b block and their backward."""

    @staticmethod
    def forward(ctx, Z, adjacency, overlook, idx, mask_rate, state, unit=False):
        lib = _lib.load()
        dev = Z.device
        b = Z.shape[0]
        Zc = Z.contiguous()
        target = torch.empty((b, b), dtype=torch.float32, device=dev)
        retain = torch.empty((b, b), dtype=torch.float32, device=dev)
        sums = torch.empty((3, b), dtype=torch.float32, device=dev)          # row_pos, row_ret, row_loss
        st = _lib.current_stream(dev)
        _lib.check(lib.sgnn_gae_targets_f32(
            adjacency.n, adjacency.nnz, _lib.ptr(adjacency.rowptr), _lib.ptr(adjacency.colidx),
            None if adjacency.unit_weights else _lib.ptr(adjacency.vals),     # all 1: no value lookups
            None if overlook is None else _lib.ptr(overlook.rowptr),
            None if overlook is None else _lib.ptr(overlook.colidx),
            b, _lib.ptr(idx), float(mask_rate), _lib.ptr(state), _lib.ptr(target), _lib.ptr(retain),
            _lib.ptr(sums[0]), _lib.ptr(sums[1]), st))
        S = Zc.matmul(Zc.t())
        G = torch.empty((b, b), dtype=torch.float32, device=dev)
        _lib.check(lib.sgnn_gae_loss_grad_f32(b, _lib.ptr(S), b, _lib.ptr(target), _lib.ptr(retain), _lib.ptr(sums[0]),
                                              _lib.ptr(sums[1]), _lib.ptr(G), _lib.ptr(sums[2]), _lib.ptr(state), st))
        loss = torch.empty((), dtype=torch.float32, device=dev)
        _lib.check(lib.sgnn_sum_f32(b, _lib.ptr(sums[2]), 1.0, _lib.ptr(loss), st))
        ctx.save_for_backward(Zc, G)
        ctx.unit = unit
        # S = Z Z^T, the mask and (for an undirected graph) the target are symmetric, hence so is
        # G: (G + G^T) Z = 2 G Z, one product without the transposed sum
        ctx.sym = bool(adjacency.symmetric and (overlook is None or overlook.symmetric))
        ctx.mark_non_differentiable(target, retain)
        ctx.set_materialize_grads(False)          # no b x b zero blocks for the two auxiliary outputs
        return loss, target, retain

    @staticmethod
    def backward(ctx, grad_loss, _gt, _gr):
        if grad_loss is None:
            return (None,) * 7
        Z, G = ctx.saved_tensors
        dZ = torch.addmm(Z, G, Z, beta=0.0, alpha=2.0) if ctx.sym else (G + G.t()).matmul(Z)
        return (dZ if ctx.unit else grad_loss * dZ), None, None, None, None, None, None


class GAELayer(_Layer):
    """One decoupled graph auto-encoder layer (SingleLayerGAE, model/SGNN.py:120-251)."""

    def __init__(self, adjacency, input_dim, embedding_dim, p, device, overlook=None):
        super().__init__(adjacency, input_dim, embedding_dim, p, device)
        self.mask_rate = p.get(LayerParam.MASK_RATE, 0.0)
        self.fused_loss = p.get("fused_loss", True)
        # entries to leave out of the reconstruction loss besides the diagonal: a
        # handle on the symmetric 0/1 overlook matrix, or None for "diagonal only"
        # (overlooked_rate 0: utils.py:47-51)
        self.overlook = overlook

    def decode(self, embedding):
        return torch.sigmoid(embedding.matmul(embedding.t()).clamp(-40.0, 40.0))     # model/SGNN.py:138-142

    def _ignore_mask(self, idx):
        b = idx.numel()
        mask = (torch.rand(b, b, device=self.device) <= self.mask_rate).float()       # model/SGNN.py:209-226
        mask = mask.maximum(mask.t())
        mask = mask - mask.diag().diag()
        if self.overlook is None:
            look = (idx[:, None] == idx[None, :]).float()       # self-loop part of the overlook matrix
        else:
            look = self.overlook.sample_block(idx, skip_self=False)
        return mask.maximum(look)

    @staticmethod
    def reconstruction_loss(recons, target, retain):
        """Positive-reweighted cross-entropy over retained entries (model/SGNN.py:682-702),
        without the host round trip of ``if num_positive > 0``."""
        num_positive = (target * retain).sum()
        ratio = retain.sum() / num_positive.clamp_min(1.0) - 1
        pos = (-ratio * target * torch.log(recons.clamp_min(1e-6)) * retain).mean()
        pos = torch.where(num_positive > 0, pos, torch.zeros_like(pos))
        neg = (-(1 - target) * torch.log((1 - recons).clamp_min(1e-6)) * retain).mean()
        return pos + neg

    def run(self, X, embedding_target=None, eta=1, train=True):
        PX = self.propagated(X)
        if train:
            def loss_fn():
                idx = self._batch(PX.shape[0])
                emb = self.embed_batch(PX, idx)
                if self.fused_loss:
                    loss = _FusedReconstruction.apply(emb, self.adjacency, self.overlook, idx, self.mask_rate,
                                                      self._rng_state, True)[0]
                else:
                    target = self.adjacency.sample_block(idx, skip_self=True)          # model/SGNN.py:167-170
                    loss = self.reconstruction_loss(self.decode(emb), target, 1 - self._ignore_mask(idx))
                if embedding_target is not None:
                    loss = loss + _squared_error(emb, embedding_target, idx, eta / emb.shape[0], unit=True)
                return loss

            self._optimise(loss_fn, lasso=self.lam)                                    # model/SGNN.py:162
        return self._finish(X, PX)


class _Stack:
    """Forward / backward sweeps over decoupled layers (StackedGNN, model/SGNN.py:470-600)."""

    def __init__(self, content, adjacency, eta, BP_count, device, labels, metric_func, logger):
        self.device = torch.device(device if device is not None else "cuda:0")
        A = adjacency.tocsr()
        A = A - _diag_of(A)                                       # model/SGNN.py:497-501
        self.adjacency = A
        self.adjacency_tensor = prop.csr_to_sparse_Tensor(A, self.device)
        self.content = torch.as_tensor(np.asarray(content), dtype=torch.float32).to(self.device)
        self.eta, self.BP_count = eta, int(BP_count)
        self.labels_host = None if labels is None else np.asarray(labels)
        self.metric_func, self.logger = metric_func, logger
        self.gnns = []

    def _forward_sweep(self, with_targets):
        inputs, x = [], self.content
        for i, g in enumerate(self.gnns):
            inputs.append(x)
            target = self.gnns[i + 1].expected_X if with_targets and i + 1 < len(self.gnns) else None
            g.set_training_direction(False, reset_backward=(i != 0))
            x = g.run(x, embedding_target=target, eta=self.eta)
        return inputs, x

    def _backward_sweep(self, inputs):
        target = None
        for i in reversed(range(len(self.gnns))):
            g = self.gnns[i]
            g.set_training_direction(i != 0)
            g.run(inputs[i], embedding_target=target, eta=self.eta)
            target = g.expected_X

    def _report(self, embedding):
        if self.labels_host is not None and self.metric_func is not None:          # model/SGNN.py:550-551
            self.metric_func(embedding.detach().cpu().numpy(), self.labels_host)

    def _train(self):
        inputs, embedding = self._forward_sweep(False)
        self._report(embedding)
        for _ in range(self.BP_count):
            self._backward_sweep(inputs)
            inputs, embedding = self._forward_sweep(True)
            self._report(embedding)
        return embedding


class StackedGNN(_Stack):
    """Unsupervised stack of auto-encoder layers (model/SGNN.py:470-564).  ``run()``
    returns the final embedding, on the device."""

    def __init__(self, content, adjacency, layers, overlooked_rates=None, eta=1, BP_count=0, device=None,
                 labels=None, metric_func=None, logger=None, rank=None):
        super().__init__(content, adjacency, eta, BP_count, device, labels, metric_func, logger)
        rates = [0.0] * len(layers) if overlooked_rates is None else list(overlooked_rates)
        dim = self.content.shape[1]
        for p, rate in zip(layers, rates):
            if p.gnn_type != LayerParam.GAE:
                raise NotImplementedError("the unsupervised stack builds auto-encoder layers only "
                                          "(model/SGNN.py:530-534 leaves the others unimplemented)")
            self.gnns.append(GAELayer(self.adjacency_tensor, dim, p.neurons, p, self.device,
                                      overlook=self._overlook(rate)))
            dim = p.neurons

    def _overlook(self, rate):
        """Symmetric random subset of the edges plus the diagonal (utils.py:14-51),
        resident on the GPU; ``None`` when only the diagonal is overlooked."""
        rate = min(max(rate, 0), 1)
        coo = self.adjacency.tocoo()
        take = int(rate * coo.nnz)
        if take == 0:
            return None
        from .norm_adj import NormAdj
        pick = torch.randperm(coo.nnz)[:take].numpy()
        # mirror images and the diagonal are added by the build; weights clamp to 1
        return NormAdj.from_coo(coo.row[pick], coo.col[pick], None, coo.shape[0], self.device,
                                symmetrize=True, clamp_one=True, strip_diag=True)

    def run(self):
        return self._train()


class SupervisedStackedGNN(_Stack):
    """Stack of decoupled layers with a label head (model/SGNN.py:628-679).  ``run()``
    returns the predicted labels of all nodes as a numpy array."""

    def __init__(self, content, adjacency, layers, training_mask, val_mask=None, labels=None,
                 overlooked_rates=None, eta=1, BP_count=0, device=None, metric_func=None, logger=None, rank=None):
        if labels is None:
            raise ValueError("labels are required")
        super().__init__(content, adjacency, eta, BP_count, device, None, metric_func, logger)
        self.labels = torch.as_tensor(np.asarray(labels)).long().to(self.device)
        self.training_mask = torch.as_tensor(np.asarray(training_mask)).to(self.device)
        n_class = int(np.unique(np.asarray(labels)).shape[0])
        dim = self.content.shape[1]
        for p in layers:
            if p.gnn_type != LayerParam.EGCN:
                raise NotImplementedError("only embedding-GCN layers (the classification runner's type) are built")
            self.gnns.append(EmbeddingLayer(self.adjacency_tensor, self.labels, self.training_mask, dim,
                                            p.neurons, n_class, p, self.device))
            dim = p.neurons

    def run(self):
        embedding = self._train()
        return self.gnns[-1].predict(embedding).cpu().numpy()


# ----------------------------------------------------------------- evaluation
def clustering_accuracy(truth, prediction):
    """Best one-to-one relabelling accuracy (metric.py:103-139; the assignment
    problem is solved with scipy instead of the munkres package)."""
    from scipy.optimize import linear_sum_assignment
    truth, prediction = np.asarray(truth).ravel(), np.asarray(prediction).ravel()
    t_ids, t = np.unique(truth, return_inverse=True)
    p_ids, q = np.unique(prediction, return_inverse=True)
    if t_ids.size != p_ids.size:
        return 0.0                                                # metric.py:109-111
    hits = np.zeros((t_ids.size, p_ids.size), dtype=np.int64)
    np.add.at(hits, (t, q), 1)
    rows, cols = linear_sum_assignment(-hits)
    return float(hits[rows, cols].sum()) / truth.size


def clustering_metrics(truth, prediction):
    from sklearn.metrics import normalized_mutual_info_score
    return clustering_accuracy(truth, prediction), float(normalized_mutual_info_score(truth, prediction))


def k_means(embedding, n_clusters, labels, replicates=1, seed=None):
    """Mean ACC / NMI of ``replicates`` k-means runs (utils.py:99-107)."""
    from sklearn.cluster import KMeans
    acc = nmi = 0.0
    for i in range(replicates):
        km = KMeans(n_clusters=n_clusters, random_state=None if seed is None else seed + i).fit(embedding)
        a, m = clustering_metrics(labels, km.predict(embedding))
        acc, nmi = acc + a, nmi + m
    return acc / replicates, nmi / replicates


def relaxed_k_means(X, n_clusters, labels, seed=None):
    """k-means on the row-normalised top singular vectors (utils.py:117-128)."""
    from sklearn.cluster import KMeans
    U, _, _ = torch.linalg.svd(X, full_matrices=False)
    ind = U[:, :n_clusters]
    ind = ind / ind.norm(dim=1, keepdim=True).clamp_min(1e-7)
    ind = ind.cpu().numpy()
    km = KMeans(n_clusters=n_clusters, random_state=seed).fit(ind)
    return clustering_metrics(labels, km.predict(ind))


def _diag_of(A):
    import scipy.sparse as sp
    n = A.shape[0]
    idx = np.arange(n)
    return sp.coo_matrix((A.diagonal(), (idx, idx)), shape=(n, n)).tocsr()
