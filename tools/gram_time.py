"""time ops.gram on the training step's shapes (C4: N = 1M)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridged_gnn_amd import ops
dev = "cuda:0"
N = 1_000_000
for p, q in ((260, 128), (128, 128), (8, 128), (288, 64), (64, 128)):
    A = torch.randn(N, p, device=dev); B = torch.randn(N, q, device=dev)
    ref = (A[:200000].double().t() @ B[:200000].double())
    got = ops.gram(A[:200000], B[:200000]).double()
    err = float((got - ref).abs().max() / ref.abs().max())
    for _ in range(3): ops.gram(A, B)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): ops.gram(A, B)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"gram p={p} q={q}: {ms:.3f} ms  {(p+q)*4*N/ms/1e6:.0f} GB/s  rel err {err:.2e}", flush=True)
