import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bridged_gnn_amd import _lib
_lib.SO_PATH = sys.argv[1]
nq, nc = int(sys.argv[2]), int(sys.argv[3])
import torch
from bridged_gnn_amd import ops, synth
dev = "cuda:0"
q = torch.from_numpy(synth.gaussian_embeddings(nq, 128, seed=0)).to(dev)
c = torch.from_numpy(synth.gaussian_embeddings(nc, 128, seed=1)).to(dev)
qn, cn = ops.l2_normalize_rows(q), ops.l2_normalize_rows(c)
idx, val, nfb = ops.cosine_topk(qn, cn, 20)
torch.cuda.synchronize()
print("done", nq, nc, int(nfb[0]))
