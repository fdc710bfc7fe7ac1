#!/usr/bin/env python3
"""Probe (GPU box, world size 1 with RCCL): partitioned forward eager vs whole-forward HIP-graph capture including the
collectives; checks the results against each other.  (Capturing only the compute BETWEEN collectives -- 7 graphs per
forward, collectives eager -- was tried and is slower than eager at every size: 0.60 vs 0.56 ms at 125k nodes.)"""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
from bridged_gnn_amd import synth
from bridged_gnn_amd.dist import PartitionedKTGNN
from bridged_gnn_amd.ktgnn import KTGNN_no_complement
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1)
hid = 128
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for n in (125_000, 250_000, 500_000, 1_000_000):
    ns = n // 2
    ei, mask = synth.bridged_graph(ns, n - ns, 6, 20, 4 * n, seed=0)
    torch.manual_seed(0)
    model = KTGNN_no_complement(hid, 2, 2, hid, use_bn=True, dim_share=hid).to(dev).eval()
    pk = PartitionedKTGNN(model, ei, mask, 0, 1, dev, always_communicate=True)
    x = torch.randn(n, hid, device=dev)[pk.owned_global].contiguous()
    with torch.no_grad():
        ref = [t.clone() for t in pk.forward(x)]
        te = bench(lambda: pk.forward(x))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outw = pk.forward(x)
        tw = bench(g.replay)
        err_w = max(float((a - b).abs().max()) for a, b in zip(ref, outw))
    print(f"N={n}: eager {te:.3f} ms | whole graph {tw:.3f} ms (max diff {err_w:.1e})", flush=True)
    del pk, g
dist.destroy_process_group()
