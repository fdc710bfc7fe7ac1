"""time the three-head classifier aggregation forward on the C4 graph"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bridged_gnn_amd import ops
dev = "cuda:0"
ei_np, mask_np = bench.c4_graph(1_000_000, 20_000_000, "local")
ei, mask = torch.from_numpy(ei_np).to(dev), torch.from_numpy(mask_np).to(dev)
csr = ops.build_dst_csr(ei, mask.shape[0], rewrite_self_loops=True)
N = mask.shape[0]
mask_u8 = mask.to(torch.uint8)
torch.manual_seed(0)
t2s = torch.randn(N, 12, device=dev); s2t = torch.randn(N, 12, device=dev)
t2s.view(N, 3, 4)[:, :, 2:] = 0; s2t.view(N, 3, 4)[:, :, 2:] = 0
a_t = torch.randn(3, 2, device=dev); a_s = torch.randn(3, 2, device=dev)
out = torch.empty(N, 12, device=dev)
for _ in range(3): ops.adaptedconv_aggregate(t2s, s2t, a_t, a_s, csr, mask_u8, 2, 0.1, heads=3, log_softmax=True, out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): ops.adaptedconv_aggregate(t2s, s2t, a_t, a_s, csr, mask_u8, 2, 0.1, heads=3, log_softmax=True, out=out)
e.record(); torch.cuda.synchronize()
print(f"heads forward {s.elapsed_time(e)/20:.3f} ms  checksum {float(out.double().sum()):.9e}", flush=True)
