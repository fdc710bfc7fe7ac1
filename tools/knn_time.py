#!/usr/bin/env python3
"""Time the cosine kNN bridge on C5 (GPU box; tuning aid) and check sampled rows against the oracle.
   BGNN_KNN_FAST_PRODUCTS=1|2|3 python tools/knn_time.py [n] [k]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bridged_gnn_amd import ops, synth
from oracle import oracle_c as OC
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda:0"
q = torch.from_numpy(synth.gaussian_embeddings(n, 128, seed=0)).to(dev)
c = torch.from_numpy(synth.gaussian_embeddings(n, 128, seed=1)).to(dev)
qn, cn = ops.l2_normalize_rows(q), ops.l2_normalize_rows(c)
ts = []
for it in range(int(os.environ.get("KNN_REPS", "6"))):
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); idx, val, nfb = ops.cosine_topk(qn, cn, k); e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e))
rows = np.arange(0, n, max(n // 96, 1))
_, ridx = OC.cosine_topk(qn[rows].cpu().numpy(), cn.cpu().numpy(), k)
ok = np.array_equal(idx[rows].cpu().numpy(), ridx)
print(f"products={os.environ.get('BGNN_KNN_FAST_PRODUCTS', '1')} n={n} k={k} ms={['%.3f' % t for t in ts]} median={np.median(ts[1:]):.3f} "
      f"exhaustive_rows={int(nfb[0])} precise_rows={int(nfb[1])} sampled_rows_exact={ok}", flush=True)
