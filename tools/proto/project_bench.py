import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sgnn_amd.stacked import _BatchProject
dev = torch.device("cuda:0")
n, F, N, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
X = torch.randn(n, F, device=dev)
W = torch.randn(F, N, device=dev, requires_grad=True)
dE = torch.randn(512, N, device=dev)
idx = torch.randperm(n, device=dev)[:512] if mode == "random" else torch.arange(512, device=dev)
for _ in range(50):
    E = _BatchProject.apply(X, idx, W, 0.2)
    E.backward(dE)
torch.cuda.synchronize()
