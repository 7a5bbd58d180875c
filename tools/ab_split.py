#!/usr/bin/env python3
"""Same-box timing of the split pass (sgnn_tilegroup_split_f32: per-feature maxima, then the fp16
planes) of two builds of the library on the benchmark shapes (run on the GPU box):
    python tools/ab_split.py <lib.so>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from sgnn_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(_root, sys.argv[1])
lib = _lib.load()
if os.environ.get("SPLIT_BLOCKS"):
    lib.sgnn_tune_set(b"tg_split_blocks", int(os.environ["SPLIT_BLOCKS"]))
if os.environ.get("SPLIT_THREADS"):
    lib.sgnn_tune_set(b"tg_split_threads", int(os.environ["SPLIT_THREADS"]))
dev = torch.device("cuda:0")
torch.manual_seed(0)
st = _lib.current_stream(dev)
for name, n, F in (("reddit", 232965, 602), ("reddit128", 232965, 128), ("products", 2449029, 100), ("rmat10m", 10_000_000, 128),
                   ("reddit41", 232965, 41)):
    X = torch.randn(n, F, device=dev)
    sr = torch.rand(n, device=dev) + 0.5
    pb = lib.sgnn_tilegroup_planes_bytes(n, F)
    planes = torch.empty(pb, dtype=torch.uint8, device=dev)
    state = torch.zeros(lib.sgnn_tilegroup_state_bytes(F) // 4, dtype=torch.int32, device=dev)
    run = lambda: _lib.check(lib.sgnn_tilegroup_split_f32(n, F, _lib.ptr(sr), _lib.ptr(X), F, _lib.ptr(planes), pb, _lib.ptr(state), st))  # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    chk = int(planes.view(torch.int32)[: pb // 4].to(torch.int64).sum().item())
    print(f"{sys.argv[1]} blocks={os.environ.get('SPLIT_BLOCKS', '-')} threads={os.environ.get('SPLIT_THREADS', '-')} {name}: max + split {e0.elapsed_time(e1) / 20:.3f} ms, checksum of the planes {chk}, "
          f"tiny {int(state[_lib.TG_STATE_TINY].item())} flag {int(state[_lib.TG_STATE_FLAG].item())}", flush=True)
    del X, planes
