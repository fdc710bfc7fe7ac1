#!/usr/bin/env python3
"""Time the AdaptedConv dense transform with the W-stationary kernel (BGNN_GEMM_WREG=1, default) vs the tiled kernel (=0)
for a few shapes (GPU box; tuning aid)."""
import os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    from bridged_gnn_amd import ops
    dev = "cuda:0"
    for (N, Din, D) in ((1_000_000, 128, 128), (1_000_000, 64, 64), (1_000_000, 128, 32), (1_000_000, 128, 2), (1_000_000, 128, 10), (1_000_000, 64, 4)):
        torch.manual_seed(0)
        x = torch.randn(N, Din, device=dev)
        mask = (torch.arange(N, device=dev) % 3 == 0).to(torch.uint8)
        delta = torch.randn(Din, device=dev)
        heads = [{"W_s": torch.randn(D, Din, device=dev) * 0.1, "W_t": torch.randn(D, Din, device=dev) * 0.1, "b_s": torch.randn(D, device=dev),
                  "b_t": torch.randn(D, device=dev), "g_s2t": torch.randn(2 * Din, device=dev) * 0.1, "g_t2s": torch.randn(2 * Din, device=dev) * 0.1}]
        pack = ops.pack_transform_heads(heads, Din)
        for _ in range(15): res = ops.adaptedconv_transform(x, mask, delta, pack)
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); res = ops.adaptedconv_transform(x, mask, delta, pack); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        h_t2s, h_s2t = res[0]
        chk = float(h_t2s.double().sum() + 2 * h_s2t.double().sum()), float(h_t2s.abs().max())
        print("WREG", os.environ.get("BGNN_GEMM_WREG", "1"), (N, Din, D), "ms med", round(float(np.median(ts)), 3), "min", round(min(ts), 3), "chk", chk, flush=True)
else:
    for w in ("0", "1"):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, BGNN_GEMM_WREG=w))
