"""Poison the caching allocator's free blocks with NaN, then run eager training steps: any NaN/inf in loss or gradients means a kernel
(or wrapper) reads memory it never wrote."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_round2 as T
DEV = "cuda:0"
data, loss_fn, a, b = T._small_training_setup(0.0)
opt = torch.optim.SGD(a.parameters(), lr=0.01)
def poison():
    ts = [torch.full((sz,), float("nan"), device=DEV) for sz in (64, 256, 1024, 4096, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24) for _ in range(6)]
    del ts
for it in range(4):
    poison()
    opt.zero_grad(set_to_none=True)
    l = loss_fn(a(data)); l.backward()
    bad = [(n, int((~torch.isfinite(p.grad)).sum())) for n, p in a.named_parameters() if not bool(torch.isfinite(p.grad).all())]
    print("iter", it, "loss", float(l.detach()), "non-finite grads:", bad, flush=True)
    opt.step()
