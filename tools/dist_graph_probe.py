#!/usr/bin/env python3
"""Probe (GPU box): can the partitioned forward -- including its RCCL collectives -- be captured into a HIP graph?
World size 1 only (the pool has one GPU per box); prints eager vs replay time for a rank-sized problem."""
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
n, hid = 125_000, 128
ns = n // 2
ei, mask = synth.bridged_graph(ns, n - ns, 6, 20, 500_000, seed=0)
torch.manual_seed(0)
model = KTGNN_no_complement(hid, 2, 2, hid, use_bn=True, dim_share=hid).to(dev).eval()
pk = PartitionedKTGNN(model, ei, mask, 0, 1, dev, always_communicate=True)
x = torch.randn(n, hid, device=dev)[pk.owned_global]
with torch.no_grad():
    for _ in range(5): ref = pk.forward(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): pk.forward(x)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 50
    print(f"eager {te*1e3:.3f} ms", flush=True)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            out = pk.forward(x)
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 50
        err = max(float((a - b).abs().max()) for a, b in zip(ref, out))
        print(f"graph replay {tg*1e3:.3f} ms, max |diff| vs eager {err:.2e}", flush=True)
    except Exception as e:
        print("capture failed:", type(e).__name__, str(e)[:300], flush=True)
dist.destroy_process_group()
