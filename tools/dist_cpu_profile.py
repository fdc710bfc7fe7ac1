import os, sys, time, cProfile, pstats
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
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
    for _ in range(10): pk.forward(x)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): pk.forward(x)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(8); st.print_callers("getenv")
dist.destroy_process_group()
