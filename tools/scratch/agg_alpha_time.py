import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bridged_gnn_amd import ops
import bench
dev = "cuda:0"; D = 128
ei, mask = bench.c4_graph(1_000_000, 20_000_000, "local")
n = mask.shape[0]
csr = ops.build_dst_csr(torch.from_numpy(ei).to(dev), n)
g = torch.Generator(device=dev).manual_seed(0)
both = torch.randn(2, n, D, device=dev, generator=g)
a1, a2 = torch.randn(D, device=dev, generator=g) * 0.3, torch.randn(D, device=dev, generator=g) * 0.3
m8 = torch.from_numpy(mask).to(dev).to(torch.uint8)
fn = lambda: ops.adaptedconv_aggregate(both[0], both[1], a1, a2, csr, m8, D, 0.1, want_alpha=True)
for _ in range(5): fn()
torch.cuda.synchronize(); ts = []
for _ in range(30):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); o, al = fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
print(f"aggregate with alpha: median {np.median(ts):.4f} ms  checksum {al.double().sum().item():.6f} {o.double().sum().item():.6e}")
