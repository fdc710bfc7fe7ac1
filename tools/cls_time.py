"""time / stamp the fused classifier-stage kernel on the C4 shapes: python tools/cls_time.py [lib.so]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridged_gnn_amd import _lib
if len(sys.argv) > 1:
    _lib.SO_PATH = sys.argv[1]
from bridged_gnn_amd import ops
dev = "cuda:0"; N, H = 1_000_000, 128
torch.manual_seed(0)
x = torch.randn(N, H, device=dev); mask = (torch.arange(N, device=dev) % 3 == 0).to(torch.uint8)
mk = lambda *s: torch.randn(*s, device=dev) * 0.1
nar = lambda: {"W_s": mk(2, H), "W_t": mk(2, H), "b_s": mk(2), "b_t": mk(2), "g_s2t": mk(2 * H), "g_t2s": mk(2 * H)}
pair, pk_t = ops.pack_transform_heads([nar(), nar()], H), ops.pack_transform_heads([nar()], H)
W, b = mk(H, H), mk(H)
sums_x = ops.domain_sums(x, mask)
t2s, s2t = torch.empty(N, 12, device=dev), torch.empty(N, 12, device=dev)
views = [(t2s[:, 4 * j:4 * j + 4], s2t[:, 4 * j:4 * j + 4]) for j in range(3)]
sums = torch.zeros(2 * H + 2, dtype=torch.float64, device=dev)
f = lambda: ops.classifier_stage(x, mask, sums_x, pair, [views[0], views[1]], W, b, sums, pk_t)
for _ in range(3): f()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); f(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
print(f"classifier_stage: median {np.median(ts):.3f} ms (min {min(ts):.3f})")
