#!/usr/bin/env python3
"""Times the W-stationary transform kernel's three modes on the C4 shapes ([1M,128] -> 2 x [1M,128] hidden transform,
plain Linear + column sums, Linear -> narrow transform) and checks a row sample of each against fp64 torch.
`python tools/transform_time.py [path/to/lib.so ...]` -- one child process per library (default: the product library)."""
import os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(so, N):
    from bridged_gnn_amd import _lib
    if so != "-":
        _lib.SO_PATH = so
    from bridged_gnn_amd import ops
    dev = "cuda:0"
    H = 128
    torch.manual_seed(0)
    x = torch.randn(N, H, device=dev)
    mask = (torch.arange(N, device=dev) % 3 == 0).to(torch.uint8)
    mk = lambda *s: torch.randn(*s, device=dev) * 0.1
    hid = {"W_s": mk(H, H), "W_t": mk(H, H), "b_s": mk(H), "b_t": mk(H), "g_s2t": mk(2 * H), "g_t2s": mk(2 * H)}
    nar = {"W_s": mk(2, H), "W_t": mk(2, H), "b_s": mk(2), "b_t": mk(2), "g_s2t": mk(2 * H), "g_t2s": mk(2 * H)}
    pk_h, pk_n = ops.pack_transform_heads([hid], H), ops.pack_transform_heads([nar], H)
    W, b = mk(H, H), mk(H)
    sums_x = ops.domain_sums(x, mask)
    out = [(torch.empty(N, H, device=dev), torch.empty(N, H, device=dev))]

    def t(fn):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
        return float(np.median(ts))
    sums = torch.zeros(2 * H + 2, dtype=torch.float64, device=dev)
    m0 = t(lambda: ops.adaptedconv_transform(x, mask, None, pk_h, out=out, sums=sums_x))
    m1 = t(lambda: ops.linear(x, W, b, relu=True, mask_u8=mask, colsum=sums))
    m2 = t(lambda: ops.linear_narrow_transform(x, W, b, mask, sums, pk_n))
    # correctness on a row sample (fp64 restatement of KTGNN.py:275-284)
    idx = torch.cat([torch.arange(0, 97, device=dev), torch.randint(0, N, (4000,), device=dev), torch.arange(N - 70, N, device=dev)])
    xd, md = x[idx].double(), mask[idx].bool()
    s = sums_x.double()
    delta = s[:H] / s[2 * H] - s[H:2 * H] / s[2 * H + 1]
    dd = delta.expand_as(xd)
    g_s = torch.tanh(torch.cat([xd, dd], 1) @ hid["g_s2t"].double())[:, None] * dd
    g_t = torch.tanh(torch.cat([xd, dd], 1) @ hid["g_t2s"].double())[:, None] * dd
    w_s2t = (xd - g_s * md[:, None]) @ hid["W_t"].double().t() + hid["b_t"].double()
    w_t2s = (xd + g_t * (~md)[:, None]) @ hid["W_s"].double().t() + hid["b_s"].double()
    h_t2s, h_s2t = ops.adaptedconv_transform(x, mask, None, pk_h, out=out, sums=sums_x)[0]
    e0 = max((h_t2s[idx].double() - w_t2s).abs().max().item(), (h_s2t[idx].double() - w_s2t).abs().max().item()) / w_t2s.abs().max().item()
    y = ops.linear(x, W, b, relu=True, mask_u8=mask, colsum=sums.zero_())
    yd = torch.relu(xd @ W.double().t() + b.double())
    e1 = (y[idx].double() - yd).abs().max().item() / yd.abs().max().item()
    cs = torch.relu(x.double() @ W.double().t() + b.double())
    e1s = ((sums[:H] - cs[mask.bool()].sum(0)).abs().max() / cs.sum(0).abs().max()).item()
    if os.environ.get("TT_TRACE"):
        torch.cuda.synchronize()
        print("   trace (cycles per tile: barrier, product, stage, epilogue, tiles) early", [round(v) for v in h_s2t[0, :5].tolist()], "late", [round(v) for v in h_s2t[0, 8:13].tolist()], "stager (barrier, load wait, sstore, gload)", [round(v) for v in h_s2t[0, 16:21].tolist()])
    print(f"{os.path.basename(so):28s} MODE0 {m0:.3f} ms (err {e0:.1e}) | MODE1 {m1:.3f} ms (err {e1:.1e}, colsum {e1s:.1e}) | MODE2 {m2:.3f} ms", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
    else:
        libs = sys.argv[1:] or ["-"]
        for so in libs:
            subprocess.run([sys.executable, __file__, "--child", so, os.environ.get("TT_N", "1000000")])
