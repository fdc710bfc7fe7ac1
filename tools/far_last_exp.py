#!/usr/bin/env python3
"""Experiment (GPU box): does moving every row's cache-unfriendly in-edges (the only edge of that row in its 4096-source
bucket) to the END of the row's edge list help the aggregation?  (Fewer steps of a wave then contain a far gather.)"""
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
E = col.numel()
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
print(f"input order: {timeit(lambda: ops.adaptedconv_aggregate(hS, hT, a1, a2, csr, m8, D, 0.1)):.3f} ms", flush=True)
for shift in (10, 12, 14):
    key = row * (1 << 22) + (col >> shift)
    uk, inv, cnt = torch.unique(key, return_inverse=True, return_counts=True)
    far = cnt[inv] == 1
    pos = torch.arange(E, device=dev)
    order = torch.argsort(row * 2 * E + far.long() * E + pos)          # stable: (row, far, original position)
    c2 = ops.DstCSR(csr.rowptr, col[order].to(torch.int32).contiguous(), None, E, n)
    out = ops.adaptedconv_aggregate(hS, hT, a1, a2, c2, m8, D, 0.1)
    t = timeit(lambda: ops.adaptedconv_aggregate(hS, hT, a1, a2, c2, m8, D, 0.1))
    print(f"bucket 2^{shift}: far edges {100.0 * float(far.float().mean()):.1f} %, far-last order {t:.3f} ms, max diff {float((out - ref).abs().max()):.1e}", flush=True)
# sorted by source index inside every row
order = torch.argsort(row * n + col)
c3 = ops.DstCSR(csr.rowptr, col[order].to(torch.int32).contiguous(), None, E, n)
print(f"sorted by source: {timeit(lambda: ops.adaptedconv_aggregate(hS, hT, a1, a2, c3, m8, D, 0.1)):.3f} ms", flush=True)
