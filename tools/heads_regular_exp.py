import sys, torch, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridged_gnn_amd import ops, synth
dev = "cuda:0"
n = 1_000_000
m8 = (torch.arange(n, device=dev) < n // 2).to(torch.uint8)
def timeit(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return float(np.median(ts))
t2s = torch.randn(n, 12, device=dev); s2t = torch.randn(n, 12, device=dev)
a1 = torch.randn(3, 2, device=dev); a2 = torch.randn(3, 2, device=dev)
for deg in (1, 4, 11, 21, 42):
    src = (torch.arange(n, device=dev).repeat_interleave(deg) + torch.randint(-500, 500, (n * deg,), device=dev)).clamp_(0, n - 1)
    rp = torch.arange(0, n * deg + 1, deg, device=dev, dtype=torch.int32)
    c = ops.DstCSR(rp, src.to(torch.int32).contiguous(), None, n * deg, n)
    t = timeit(lambda: ops.adaptedconv_aggregate(t2s, s2t, a1, a2, c, m8, 2, 0.1, heads=3))
    hS = torch.randn(n, 128, device=dev); hT = torch.randn(n, 128, device=dev); b1 = torch.randn(128, device=dev) * 0.1
    tw = timeit(lambda: ops.adaptedconv_aggregate(hS, hT, b1, b1, c, m8, 128, 0.1))
    del hS, hT
    print(f"heads=3 regular deg {deg}: {t:.3f} ms ({n*deg/t/1e6:.1f} G edges/s) | wide D=128: {tw:.3f} ms ({n*deg/tw/1e6:.1f} G edges/s)", flush=True)

# C4-style graphs: how much of the gap to the regular graph is the far edges, how much the degree mix?
ns = n // 2
for name, kw in (("C4 local p=0.9", dict(p_local=0.9)), ("C4 all-local p=1.0", dict(p_local=1.0)), ("C4 uniform", dict(p_local=0.0))):
    ei, mask = synth.bridged_graph(ns, n - ns, 6, 20, 4_000_000, seed=0, **kw)
    csr = ops.build_dst_csr(torch.from_numpy(ei).to(dev), n)
    mm = torch.from_numpy(mask).to(dev).to(torch.uint8)
    t = timeit(lambda: ops.adaptedconv_aggregate(t2s, s2t, a1, a2, csr, mm, 2, 0.1, heads=3))
    hS = torch.randn(n, 128, device=dev); hT = torch.randn(n, 128, device=dev); b1 = torch.randn(128, device=dev) * 0.1
    tw = timeit(lambda: ops.adaptedconv_aggregate(hS, hT, b1, b1, csr, mm, 128, 0.1))
    print(f"heads=3 {name}: {t:.3f} ms ({csr.num_edges/t/1e6:.1f} G edges/s) | wide D=128: {tw:.3f} ms ({csr.num_edges/tw/1e6:.1f} G edges/s)", flush=True)
    del hS, hT
