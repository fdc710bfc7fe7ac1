import sys, numpy as np, torch
sys.path.insert(0,'.')
from bridged_gnn_amd import ops, synth
from bridged_gnn_amd.ktgnn import AdaptedConv
dev='cuda:0'
n=40000
ei, mask = synth.bridged_graph(n//2, n-n//2, k_within=4, k_cross=8, n_extra=20000, cluster=256, seed=3)
m=torch.from_numpy(mask).to(dev)
csr=ops.build_dst_csr(torch.from_numpy(ei).to(dev), n, rewrite_self_loops=True)
need=csr.tile_need(m.to(torch.uint8))
print("need:", None if need is None else torch.bincount(need, minlength=4).tolist())
torch.manual_seed(0)
conv=AdaptedConv(128,128).to(dev).eval()
x=torch.randn(n,128,device=dev)
with torch.no_grad():
    a=conv(x,None,central_mask=m,csr=csr)
    import os
    csr._tile_need=(csr._tile_need[0],None,csr._tile_need[2])   # force "all needed"
    b=conv(x,None,central_mask=m,csr=csr)
print("equal:", torch.equal(a,b), float((a-b).abs().max()))
