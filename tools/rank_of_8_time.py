"""Compute side of ONE rank of an 8-way partition of C4, on one GPU: the real PartitionPlan of rank r (interior /
boundary rows, send lists, halo numbering), the collectives replaced by local stand-ins of the same shape (halo rows
filled by a device copy, all-reduce = identity).  Numbers are the per-rank GPU work the 8-GPU run cannot go below;
outputs are NOT the model's (the halo holds stand-in rows)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridged_gnn_amd import synth, dist as D
from bridged_gnn_amd.ktgnn import KTGNN_no_complement

world = int(os.environ.get("WORLD", "8")); rank = int(os.environ.get("RANKSIM", "0"))
n, e = 1_000_000, 20_000_000
dev = torch.device("cuda:0")
ns = n // 2
ei, mask = synth.bridged_graph(ns, n - ns, 6, 20, e - 6 * n - 20 * (n - ns), cluster=1024, p_local=0.9, seed=0)   # bench.py's C4
torch.manual_seed(0)
model = KTGNN_no_complement(128, 2, 2, 128, use_bn=True, dim_share=128).to(dev).eval()


class FakeHalo(D.HaloExchange):
    def start(self, big):
        p = self.plan
        from bridged_gnn_amd import ops
        send = ops.gather_rows(big, self.send_rows)
        recv = big[2 * p.n_local: 2 * p.n_local + p.n_halo]
        k = min(send.shape[0], recv.shape[0])
        recv[:k].copy_(send[:k])                       # stand-in payload of the right size
        self._keep, self._work = send, None

    def exchange_rows(self, rows):                     # input-feature halo (once): stand-in rows of the right shape
        p = self.plan
        recv = torch.zeros(p.n_halo, rows.shape[1], dtype=rows.dtype, device=rows.device)
        k = min(rows.shape[0], p.n_halo)
        recv[:k] = rows[:k]
        return recv


pk = D.PartitionedKTGNN(model, ei, mask, rank, world, dev, always_communicate=False,
                        cache_input_halo=os.environ.get("CACHE_HALO", "1") != "0")
pk.halo = FakeHalo(pk.plan, dev, None)
pk.world = 1
pk.always = True                                        # ... but the code path of world > 1 (two-pass classifier stage, one batched all-reduce):
pk._all_reduce = lambda t: t                             # the all-reduce itself is the identity stand-in
print("plan", pk.plan.summary(), "send rows", int(pk.halo.send_rows.numel()), flush=True)
x = torch.randn(pk.plan.n_local, 128, device=dev)
with torch.no_grad():
    for _ in range(5): pk.forward(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): pk.forward(x)
    torch.cuda.synchronize()
    print("eager ms", (time.perf_counter() - t0) / 50 * 1e3, flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        pk.forward(x)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    print("graph ms", (time.perf_counter() - t0) / 50 * 1e3, flush=True)
