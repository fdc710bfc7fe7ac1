#!/usr/bin/env python3
"""Ablation of the fused linear -> narrow transform kernel (transform_wreg_kernel<.., MODE 2>): variants built with
-DM2X=k into tools/exp_libs/lib_m2x_k.so (1: no second-stage MFMAs, 2: no ds_adds, 3: no flush/second barrier,
4: no ctile read-back for the column sums) against bgnn_linear_f32 (MODE 1) on 1M x 128."""
import os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    from bridged_gnn_amd import _lib
    _lib.SO_PATH = sys.argv[1]
    from bridged_gnn_amd import ops
    dev = "cuda:0"
    N, H = 1_000_000, 128
    torch.manual_seed(0)
    x = torch.randn(N, H, device=dev)
    mask = (torch.arange(N, device=dev) % 3 == 0).to(torch.uint8)
    W = torch.randn(H, H, device=dev) * 0.1; b = torch.randn(H, device=dev)
    head = {"W_s": torch.randn(2, H, device=dev) * 0.1, "W_t": torch.randn(2, H, device=dev) * 0.1, "b_s": torch.randn(2, device=dev),
            "b_t": torch.randn(2, device=dev), "g_s2t": torch.randn(2 * H, device=dev) * 0.1, "g_t2s": torch.randn(2 * H, device=dev) * 0.1}
    pack = ops.pack_transform_heads([head], H)

    def t(fn):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        return float(np.median(ts))
    sums = torch.zeros(2 * H + 2, dtype=torch.float64, device=dev)
    a = t(lambda: ops.linear(x, W, b, relu=True, mask_u8=mask, colsum=sums))
    c = t(lambda: ops.linear_narrow_transform(x, W, b, mask, sums, pack))
    print(os.path.basename(sys.argv[1]), "linear (MODE 1) ms", round(a, 3), "| fused stage A (MODE 2) ms", round(c, 3), flush=True)
else:
    for k in range(1):
        subprocess.run([sys.executable, __file__, os.path.join(ROOT, "bridged_gnn_amd", "csrc", "libbgnn_hip.so")])
