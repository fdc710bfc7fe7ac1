#!/usr/bin/env python3
"""Times the hidden conv's aggregation BACKWARD (pull form, D = 128: pass A by destination + pass B by source, what the C4 training step issues)
on the C4 graph.  `python tools/agg_bwd_time.py [--graph local|uniform] [path/to/lib.so ...]` -- one child process per library
(default: the product library); BGNN_AGG_FAST=0 in the environment times the general kernel."""
import os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(so, graph):
    from bridged_gnn_amd import _lib
    if so != "-":
        _lib.SO_PATH = so
    from bridged_gnn_amd import ops
    import bench
    dev = "cuda:0"
    D = 128
    ei, mask = bench.c4_graph(1_000_000, 20_000_000, graph)
    n = mask.shape[0]
    csr = ops.build_dst_csr(torch.from_numpy(ei).to(dev), n)
    g = torch.Generator(device=dev).manual_seed(0)
    both = torch.randn(2, n, D, device=dev, generator=g)
    a1, a2 = torch.randn(D, device=dev, generator=g) * 0.3, torch.randn(D, device=dev, generator=g) * 0.3
    sc, sh = torch.rand(D, device=dev, generator=g) + 0.5, torch.randn(D, device=dev, generator=g)
    m8 = torch.from_numpy(mask).to(dev).to(torch.uint8)
    out, alpha = ops.adaptedconv_aggregate(both[0], both[1], a1, a2, csr, m8, D, 0.1, want_alpha=True)
    gr = torch.randn(n, D, device=dev, generator=g)
    csr.transposed()
    res = []
    def fn():
        res[:] = ops.adaptedconv_aggregate_bwd(both[0], both[1], a1, a2, csr, m8, D, out, alpha, gr, 0.1)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    print(f"{os.path.basename(so):28s} {graph}: median {np.median(ts):.4f} ms  min {min(ts):.4f}  checksum {res[0].double().sum().item():.6e} {res[1].double().sum().item():.6e} {res[2].double().sum().item():.6e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3])
        sys.exit(0)
    args = sys.argv[1:]
    graph = "local"
    if args[:1] == ["--graph"]:
        graph, args = args[1], args[2:]
    for so in (args or ["-"]):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", so, graph], check=True)
