"""The fused pieces of an optimiser iteration (Adam update, label-head cross-entropy,
embedding-target squared error) against the stock torch operators they replace."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu


def _params(dev, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    shapes = [(602, 128), (1, 128), (602, 602), (128, 41), (3,), (1025,)]
    return [torch.nn.Parameter(torch.randn(*s, device=dev, generator=g) * 0.1) for s in shapes], g


@pytest.mark.parametrize("l1", [0.0, 1e-3])
def test_fused_adam_follows_torch_adam(gpu_device, l1):
    from sgnn_amd.stacked import _FusedAdam
    mine, g = _params(gpu_device, 1)
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    split = 3
    fused = _FusedAdam([dict(params=mine[:split], weight_decay=2e-3, l1_decay=l1),
                        dict(params=mine[split:], weight_decay=0.0)], 1e-2, gpu_device)
    stock = torch.optim.Adam([dict(params=ref[:split], weight_decay=2e-3),
                              dict(params=ref[split:], weight_decay=0.0)], lr=1e-2)
    for it in range(40):
        for k, (a, b) in enumerate(zip(mine, ref)):
            if k == 4 and it % 2:                    # a parameter without a gradient is left alone
                a.grad = b.grad = None
                continue
            grad = torch.randn(a.shape, device=gpu_device, generator=g)
            a.grad = grad.clone()
            b.grad = grad.clone() + (l1 * torch.sign(b.detach()) if k < split else 0)
        fused.step()
        stock.step()
    assert int(fused.state[0].item()) == 40 and int(fused.state[1:].abs().sum().item()) == 0
    for a, b in zip(mine, ref):
        if a.shape == (3,):
            continue                                  # stepped on its own schedule above: torch keeps a per-tensor count
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)


def test_fused_adam_counts_steps_under_graph_replay(gpu_device):
    from sgnn_amd.stacked import _FusedAdam
    p = torch.nn.Parameter(torch.ones(5000, device=gpu_device))
    q = torch.nn.Parameter(torch.ones(5000, device=gpu_device))
    p.grad = torch.full_like(p, 0.5)
    q.grad = torch.full_like(q, 0.5)
    fused = _FusedAdam([dict(params=[p])], 1e-2, gpu_device)
    stock = torch.optim.Adam([q], lr=1e-2)
    side = torch.cuda.Stream(gpu_device)
    side.wait_stream(torch.cuda.current_stream(gpu_device))
    with torch.cuda.stream(side):
        fused.step()
    torch.cuda.current_stream(gpu_device).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fused.step()
    for _ in range(9):                                           # capturing records the step without running it
        graph.replay()
    for _ in range(10):
        stock.step()
    torch.cuda.synchronize()
    assert int(fused.state[0].item()) == 10
    np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6)


@pytest.mark.parametrize("B,C", [(1, 1), (17, 41), (512, 41), (512, 64), (1024, 3), (300, 65), (64, 200), (1025, 41), (20000, 41)])
@pytest.mark.parametrize("indexed", [False, True])
def test_softmax_xent_matches_torch(gpu_device, B, C, indexed):
    from sgnn_amd.stacked import _cross_entropy
    g = torch.Generator(device=gpu_device); g.manual_seed(B * 1000 + C)
    z = (torch.randn(B, C, device=gpu_device, generator=g) * 4).requires_grad_()
    z2 = z.detach().clone().requires_grad_()
    n_lab = 3 * B if indexed else B
    labels = torch.randint(0, C, (n_lab,), device=gpu_device, generator=g)
    idx = torch.randperm(n_lab, device=gpu_device, generator=g)[:B] if indexed else None
    mine = _cross_entropy(z, labels, idx)
    ref = Fn.cross_entropy(z2, labels[idx] if indexed else labels)
    (3.0 * mine).backward()
    (3.0 * ref).backward()
    assert abs(mine.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    np.testing.assert_allclose(z.grad.cpu().numpy(), z2.grad.cpu().numpy(), rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("B,D", [(1, 1), (512, 128), (512, 64), (100, 130), (30001, 64), (1024, 128), (7, 4)])
@pytest.mark.parametrize("indexed", [False, True])
@pytest.mark.parametrize("strided", [False, True])
def test_squared_error_matches_torch(gpu_device, B, D, indexed, strided):
    from sgnn_amd.stacked import _squared_error
    g = torch.Generator(device=gpu_device); g.manual_seed(B + D)
    a = torch.randn(B, D, device=gpu_device, generator=g).requires_grad_()
    a2 = a.detach().clone().requires_grad_()
    rows = 4 * B if indexed else B
    wide = torch.randn(rows, D + 6, device=gpu_device, generator=g)
    # strided rows (generic kernels) or a compact block (the one-launch kernel when it is batch-sized)
    target = wide[:, 3:3 + D] if strided else wide[:, 3:3 + D].contiguous()
    idx = torch.randint(0, rows, (B,), device=gpu_device, generator=g) if indexed else None
    scale = 1000.0 / B
    mine = _squared_error(a, target, idx, scale)
    ref = scale * (a2 - (target[idx] if indexed else target)).pow(2).sum()
    (0.5 * mine).backward()
    (0.5 * ref).backward()
    assert abs(mine.item() - ref.item()) <= 1e-5 * abs(ref.item())
    np.testing.assert_allclose(a.grad.cpu().numpy(), a2.grad.cpu().numpy(), rtol=1e-6, atol=1e-8)


def test_identity_projection_is_skipped_exactly(gpu_device):
    from sgnn_amd.stacked import EmbeddingLayer, LayerParam, get_activation
    from sgnn_amd.norm_adj import NormAdj
    n, F = 300, 70
    rng = np.random.default_rng(0)
    r, c = rng.integers(0, n, 2000), rng.integers(0, n, 2000)
    adj = NormAdj.from_coo(r, c, None, n, gpu_device)
    p = LayerParam(16, inner_act=get_activation("leaky_relu=0.2"), act=get_activation("linear"))
    layer = EmbeddingLayer(adj, torch.zeros(n, dtype=torch.long, device=gpu_device),
                           torch.ones(n, dtype=torch.bool, device=gpu_device), F, 16, 2, p, gpu_device)
    X = torch.randn(n, F, device=gpu_device)
    assert layer._u_identity
    skipped = layer(X)
    layer._u_identity = False
    assert torch.equal(skipped, layer(X))
    layer.set_training_direction(True)
    assert not layer._u_identity
    layer.set_training_direction(False)                          # resets U to the identity
    assert layer._u_identity and torch.equal(layer.U, torch.eye(F, device=gpu_device))


def test_train_step_entry_points_reject_bad_arguments(gpu_device):
    from sgnn_amd import _lib
    lib = _lib.load()
    state = torch.zeros(_lib.ADAM_STATE_WORDS, dtype=torch.int64, device=gpu_device)
    st = _lib.current_stream(gpu_device)
    arr = (ctypes.c_void_p * 17)()
    sizes = (ctypes.c_int64 * 17)()
    assert lib.sgnn_adam_step_f32(17, arr, arr, arr, arr, sizes, None, None, 1e-3, 0.9, 0.999, 1e-8,
                                  _lib.ptr(state), st) != 0
    assert lib.sgnn_adam_step_f32(1, arr, arr, arr, arr, sizes, None, None, 1e-3, 1.0, 0.999, 1e-8,
                                  _lib.ptr(state), st) != 0
    assert lib.sgnn_adam_step_f32(0, None, None, None, None, None, None, None, 1e-3, 0.9, 0.999, 1e-8,
                                  _lib.ptr(state), st) == 0          # nothing to do
    z = torch.zeros(4, 4, device=gpu_device)
    lab = torch.zeros(4, dtype=torch.int64, device=gpu_device)
    loss = torch.zeros((), device=gpu_device)
    ws = torch.zeros(4, device=gpu_device)
    w = (_lib.ptr(ws), 16, st)
    assert lib.sgnn_loss_workspace_bytes(4) == 16
    assert lib.sgnn_softmax_xent_f32(4, 4, _lib.ptr(z), 3, _lib.ptr(lab), 4, None, _lib.ptr(loss), _lib.ptr(z), 4, *w) != 0
    assert lib.sgnn_softmax_xent_f32(4, 4, _lib.ptr(z), 4, _lib.ptr(lab), 3, None, _lib.ptr(loss), _lib.ptr(z), 4, *w) != 0
    assert lib.sgnn_softmax_xent_f32(4, 4, _lib.ptr(z), 4, _lib.ptr(lab), 4, None, _lib.ptr(loss), _lib.ptr(z), 4,
                                     _lib.ptr(ws), 12, st) != 0                            # workspace too small
    assert lib.sgnn_sq_err_f32(4, 4, _lib.ptr(z), 4, _lib.ptr(z), 4, 3, None, 1.0, _lib.ptr(loss), _lib.ptr(z), 4, *w) != 0
    assert lib.sgnn_sq_err_f32(4, 4, _lib.ptr(z), 4, _lib.ptr(z), 4, 4, None, 1.0, _lib.ptr(loss), _lib.ptr(z), 4,
                               None, 16, st) != 0                                           # no workspace
    assert lib.sgnn_batch_project_workspace_bytes(512, 602, 128, 0) == 19 * 8 * 2 * 16384
    assert lib.sgnn_batch_project_f32(8, 64, _lib.ptr(z), 64, 4, _lib.ptr(lab), 0.2, _lib.ptr(z), 4, 4, _lib.ptr(z), 4,
                                      None, 0, st) != 0                                     # two chunks, no workspace
    # one chunk (N <= 32) with a derivative mask still runs at depth 2: half the workspace is refused
    need = lib.sgnn_batch_project_workspace_bytes(4, 4, 4, 2)
    assert need == 2 * 16384
    big = torch.zeros(need // 4, device=gpu_device)
    assert lib.sgnn_batch_project_input_grad_f32(4, 4, 4, _lib.ptr(z), 4, _lib.ptr(z), 4, _lib.ptr(z), 4, 0.2,
                                                 _lib.ptr(z), 4, _lib.ptr(big), need // 2, st) != 0
    assert lib.sgnn_batch_project_input_grad_f32(4, 4, 4, _lib.ptr(z), 4, _lib.ptr(z), 4, _lib.ptr(z), 4, 0.2,
                                                 _lib.ptr(z), 4, _lib.ptr(big), need, st) == 0
    torch.cuda.synchronize()
    assert int(state[0].item()) == 0


@pytest.mark.parametrize("b,F,N", [(512, 602, 128), (512, 128, 64), (1, 1, 1), (33, 65, 31), (100, 40, 200)])
@pytest.mark.parametrize("slope", [1.0, 0.0, 0.2])
def test_batch_projection_matches_torch(gpu_device, b, F, N, slope):
    from sgnn_amd.stacked import _BatchProject
    g = torch.Generator(device=gpu_device); g.manual_seed(b + F + N)
    n = 3 * b + 5
    wide = torch.randn(n, F + 3, device=gpu_device, generator=g)
    X = wide[:, 1:1 + F]                                          # strided rows
    idx = torch.randint(0, n, (b,), device=gpu_device, generator=g)
    W = (torch.randn(F, N, device=gpu_device, generator=g) * 0.1).requires_grad_()
    W2 = W.detach().clone().requires_grad_()
    dE = torch.randn(b, N, device=gpu_device, generator=g)
    E = _BatchProject.apply(X, idx, W, slope)
    ref = Fn.leaky_relu(X[idx].double(), slope).matmul(W2.double())
    E.backward(dE)
    ref.backward(dE.double())
    scale = (Fn.leaky_relu(X[idx].double(), slope).abs().matmul(W2.detach().double().abs())).clamp_min(1e-30)
    assert ((E.double() - ref).abs() / scale).max().item() <= 1e-5
    gscale = (Fn.leaky_relu(X[idx].double(), slope).abs().t().matmul(dE.double().abs())).clamp_min(1e-30)
    assert ((W.grad.double() - W2.grad.double()).abs() / gscale).max().item() <= 1e-5


def test_embed_batch_equals_the_unfused_layer(gpu_device):
    from sgnn_amd.norm_adj import NormAdj
    from sgnn_amd.stacked import EmbeddingLayer, LayerParam, get_activation
    n, F = 400, 70
    rng = np.random.default_rng(0)
    adj = NormAdj.from_coo(rng.integers(0, n, 2000), rng.integers(0, n, 2000), None, n, gpu_device)
    for inner, outer in (("leaky_relu=0.2", "linear"), ("relu", "tanh"), ("tanh", "linear")):
        p = LayerParam(16, inner_act=get_activation(inner), act=get_activation(outer))
        layer = EmbeddingLayer(adj, torch.zeros(n, dtype=torch.long, device=gpu_device),
                               torch.ones(n, dtype=torch.bool, device=gpu_device), F, 16, 2, p, gpu_device)
        X = torch.randn(n, F, device=gpu_device)
        idx = torch.randperm(n, device=gpu_device)[:64]
        fused = layer.embed_batch(X, idx)
        plain = layer(X[idx])
        np.testing.assert_allclose(fused.detach().cpu().numpy(), plain.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
        fused.sum().backward()
        g1 = layer.W.grad.clone(); layer.W.grad = None
        plain.sum().backward()
        np.testing.assert_allclose(g1.cpu().numpy(), layer.W.grad.cpu().numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,D,C", [(512, 128, 41), (512, 64, 41), (5, 3, 2), (100, 200, 64), (1000, 128, 7), (9, 1, 1)])
@pytest.mark.parametrize("indexed", [False, True])
def test_label_head_matches_torch(gpu_device, B, D, C, indexed):
    from sgnn_amd.stacked import _head_cross_entropy
    g = torch.Generator(device=gpu_device); g.manual_seed(B + D + C)
    E = torch.randn(B, D, device=gpu_device, generator=g).requires_grad_()
    Wt = (torch.randn(D, C, device=gpu_device, generator=g) * 0.5).requires_grad_()
    E2, Wt2 = E.detach().clone().requires_grad_(), Wt.detach().clone().requires_grad_()
    n_lab = 3 * B if indexed else B
    labels = torch.randint(0, C, (n_lab,), device=gpu_device, generator=g)
    idx = torch.randperm(n_lab, device=gpu_device, generator=g)[:B] if indexed else None
    mine = _head_cross_entropy(E, Wt, labels, idx)
    ref = Fn.cross_entropy(E2.double().matmul(Wt2.double()), labels[idx] if indexed else labels)
    (2.0 * mine).backward()
    (2.0 * ref).backward()
    assert abs(mine.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    for got, want in ((E.grad, E2.grad), (Wt.grad, Wt2.grad)):
        err = (got.double() - want.double()).norm() / want.double().norm().clamp_min(1e-30)
        assert float(err) <= 1e-5, float(err)
    # the unfused route (wide heads) gives the same thing
    wide = torch.randn(B, 300, device=gpu_device, generator=g)
    Ww = torch.randn(300, C, device=gpu_device, generator=g) * 0.1
    a = _head_cross_entropy(wide, Ww, labels, idx)
    b = Fn.cross_entropy(wide.matmul(Ww), labels[idx] if indexed else labels)
    assert abs(a.item() - b.item()) <= 1e-5 * max(1.0, abs(b.item()))


@pytest.mark.parametrize("b,F,N", [(512, 602, 128), (512, 128, 64), (1, 1, 1), (33, 65, 31), (100, 40, 200), (70, 300, 20)])
@pytest.mark.parametrize("slope", [1.0, 0.0, 0.2])
def test_dense_projection_matches_torch(gpu_device, b, F, N, slope):
    from sgnn_amd.stacked import _DenseProject
    g = torch.Generator(device=gpu_device); g.manual_seed(7 * b + F + N)
    A = torch.randn(b, F, device=gpu_device, generator=g).requires_grad_()
    W = (torch.randn(F, N, device=gpu_device, generator=g) * 0.1).requires_grad_()
    A2, W2 = A.detach().clone().double().requires_grad_(), W.detach().clone().double().requires_grad_()
    dE = torch.randn(b, N, device=gpu_device, generator=g)
    E = _DenseProject.apply(A, W, slope)
    ref = Fn.leaky_relu(A2, slope).matmul(W2)
    E.backward(dE)
    ref.backward(dE.double())
    H = Fn.leaky_relu(A2.detach(), slope).abs()
    scale = H.matmul(W2.detach().abs()).clamp_min(1e-30)
    assert ((E.double() - ref).abs() / scale).max().item() <= 1e-5
    gW = H.t().matmul(dE.double().abs()).clamp_min(1e-30)
    assert ((W.grad.double() - W2.grad).abs() / gW).max().item() <= 1e-5
    gA = dE.double().abs().matmul(W2.detach().abs().t()).clamp_min(1e-30)
    assert ((A.grad.double() - A2.grad).abs() / gA).max().item() <= 1e-5
