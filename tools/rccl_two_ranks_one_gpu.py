"""Probe: can two RCCL ranks share the single GPU of the test box?  (If yes, the real all_to_all_single path can
be exercised at world size 2 before the driver's multi-GPU run.)"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl")
from bridged_gnn_amd import synth
from bridged_gnn_amd.data import Data
from bridged_gnn_amd.dist import PartitionedKTGNN
from bridged_gnn_amd.ktgnn import KTGNN_no_complement
ei, mask = synth.bridged_graph(3000, 2000, 4, 8, 6000, cluster=128, seed=4)
torch.manual_seed(0)
m = KTGNN_no_complement(64, 3, 2, 64, use_bn=True, dim_share=64).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(5000, 64, device=dev, generator=g)
with torch.no_grad():
    ref = m(Data(x=x, edge_index=torch.from_numpy(ei).to(dev), central_mask=torch.from_numpy(mask).to(dev)))[:3]
pk = PartitionedKTGNN(m, ei, mask, rank, world, dev)
out = pk.forward(x[pk.owned_global])
ok = all(torch.allclose(a, b[pk.owned_global], rtol=1e-5, atol=1e-5) for a, b in zip(out, ref))
print(f"rank {rank}: n_local {pk.plan.n_local} halo {pk.plan.n_halo} match={ok}", flush=True)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
