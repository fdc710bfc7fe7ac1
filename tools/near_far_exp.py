#!/usr/bin/env python3
"""Experiment (GPU box): aggregate the L2-friendly ("near": |src - dst| <= T) in-edges of every row first and the far ones
in a second launch that resumes the online-softmax state (part 1 / part 2 of the ABI) -- does keeping the far gathers'
HBM latency out of the near pass beat one pass over all edges?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bridged_gnn_amd import ops, synth
dev = "cuda:0"
n, D = 1_000_000, 128
ns = n // 2
ei, mask = synth.bridged_graph(ns, n - ns, 6, 20, 4_000_000, p_local=0.9, seed=0)
csr = ops.build_dst_csr(torch.from_numpy(ei).to(dev), n)
m8 = torch.from_numpy(mask).to(dev).to(torch.uint8)
hS = torch.randn(n, D, device=dev); hT = torch.randn(n, D, device=dev)
a1 = torch.randn(D, device=dev) * 0.1; a2 = torch.randn(D, device=dev) * 0.1
rowptr, col = csr.rowptr.long(), csr.col.long()
row = torch.repeat_interleave(torch.arange(n, device=dev), rowptr[1:] - rowptr[:-1])
def timeit(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return float(np.median(ts))
ref = ops.adaptedconv_aggregate(hS, hT, a1, a2, csr, m8, D, 0.1)
t_one = timeit(lambda: ops.adaptedconv_aggregate(hS, hT, a1, a2, csr, m8, D, 0.1))
print(f"one pass: {t_one:.3f} ms", flush=True)
for T in (2048, 8192, 65536):
    # "near" also needs the SAME table: a source in the other domain's index range is far by construction
    far = (col - row).abs() > T
    def sub(sel):
        r, c = row[sel], col[sel]
        rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rp[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
        return ops.DstCSR(rp.to(torch.int32).contiguous(), c.to(torch.int32).contiguous(), None, int(c.numel()), n)
    cn, cf = sub(~far), sub(far)
    out = torch.empty(n, D, device=dev)
    st = torch.empty(n, 2, device=dev)
    def two():
        ops.adaptedconv_aggregate(hS, hT, a1, a2, cn, m8, D, 0.1, out=out, state_ms=st, part=1)
        ops.adaptedconv_aggregate(hS, hT, a1, a2, cf, m8, D, 0.1, out=out, state_ms=st, part=2)
    two(); torch.cuda.synchronize()
    err = float((out - ref).abs().max())
    t_two = timeit(two)
    t_near = timeit(lambda: ops.adaptedconv_aggregate(hS, hT, a1, a2, cn, m8, D, 0.1, out=out, state_ms=st, part=1))
    print(f"T={T}: far edges {int(far.sum())} ({100.0 * float(far.float().mean()):.1f} %), near pass {t_near:.3f} ms, both passes {t_two:.3f} ms, max diff {err:.1e}", flush=True)

# fixed per-row cost of the kernel: the same 1M rows with 1, 4 and 11 in-edges each (local windows)
for deg in (1, 4, 11, 21, 42):
    src = (torch.arange(n, device=dev).repeat_interleave(deg) + torch.randint(-500, 500, (n * deg,), device=dev)).clamp_(0, n - 1)
    r = torch.arange(n, device=dev).repeat_interleave(deg)
    rp = torch.arange(0, n * deg + 1, deg, device=dev, dtype=torch.int32)
    c = ops.DstCSR(rp, src.to(torch.int32).contiguous(), None, n * deg, n)
    t = timeit(lambda: ops.adaptedconv_aggregate(hS, hT, a1, a2, c, m8, D, 0.1))
    print(f"deg {deg}: {t:.3f} ms  ({n * deg / t / 1e6:.1f} G edges/s)", flush=True)
