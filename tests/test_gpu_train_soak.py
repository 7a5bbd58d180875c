"""Random shapes through the batch-projection, dense-projection and label-head kernels against
float64 torch: every product within 1e-5 of the magnitude its terms sum to, the head's loss and
gradients within 1e-5 normwise."""
import random

import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu


def test_random_shapes_against_float64(gpu_device):
    from sgnn_amd.stacked import _BatchProject, _DenseProject, _head_cross_entropy
    dev = gpu_device
    g = torch.Generator(device=dev); g.manual_seed(0)
    random.seed(0)
    worst = 0.0
    for it in range(120):
        b, F, N = random.choice([1, 2, 31, 64, 65, 200, 512, 1000]), random.randint(1, 700), random.randint(1, 300)
        slope = random.choice([1.0, 0.0, 0.2])
        n = b * 2 + 3
        X = torch.randn(n, F, device=dev, generator=g)
        idx = torch.randint(0, n, (b,), device=dev, generator=g)
        W = (torch.randn(F, N, device=dev, generator=g) * 0.1).requires_grad_()
        dE = torch.randn(b, N, device=dev, generator=g)
        E = _BatchProject.apply(X, idx, W, slope); E.backward(dE)
        Wd = W.detach().double().requires_grad_()
        H = Fn.leaky_relu(X[idx].double(), slope)
        ref = H.matmul(Wd); ref.backward(dE.double())
        e1 = ((E.double() - ref).abs() / H.abs().matmul(Wd.detach().abs()).clamp_min(1e-30)).max().item()
        e2 = ((W.grad.double() - Wd.grad).abs() / H.abs().t().matmul(dE.double().abs()).clamp_min(1e-30)).max().item()
        A = torch.randn(b, F, device=dev, generator=g).requires_grad_()
        W.grad = None
        E = _DenseProject.apply(A, W, slope); E.backward(dE)
        Ad = A.detach().double().requires_grad_(); Wd = W.detach().double().requires_grad_()
        ref = Fn.leaky_relu(Ad, slope).matmul(Wd); ref.backward(dE.double())
        e3 = ((A.grad.double() - Ad.grad).abs() / dE.double().abs().matmul(Wd.detach().abs().t()).clamp_min(1e-30)).max().item()
        e4 = ((W.grad.double() - Wd.grad).abs() / Fn.leaky_relu(Ad.detach(), slope).abs().t().matmul(dE.double().abs()).clamp_min(1e-30)).max().item()
        # head
        D, C = random.randint(1, 256), random.randint(1, 64)
        Eh = torch.randn(b, D, device=dev, generator=g).requires_grad_()
        Wt = (torch.randn(D, C, device=dev, generator=g) * 0.3).requires_grad_()
        lab = torch.randint(0, C, (3 * b,), device=dev, generator=g)
        ii = torch.randint(0, 3 * b, (b,), device=dev, generator=g)
        l = _head_cross_entropy(Eh, Wt, lab, ii); l.backward()
        Ed, Wtd = Eh.detach().double().requires_grad_(), Wt.detach().double().requires_grad_()
        lr = Fn.cross_entropy(Ed.matmul(Wtd), lab[ii]); lr.backward()
        e5 = abs(l.item() - lr.item()) / max(1.0, abs(lr.item()))
        e6 = ((Eh.grad.double() - Ed.grad).norm() / Ed.grad.norm().clamp_min(1e-30)).item()
        e7 = ((Wt.grad.double() - Wtd.grad).norm() / Wtd.grad.norm().clamp_min(1e-30)).item()
        m = max(e1, e2, e3, e4, e5, e6, e7)
        worst = max(worst, m)
        assert m <= 1e-5, (it, b, F, N, slope, D, C, [f"{v:.2e}" for v in (e1, e2, e3, e4, e5, e6, e7)])
