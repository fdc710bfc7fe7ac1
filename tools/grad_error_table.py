"""per-tensor relative gradient error of AdaptedConv (HIP backward) vs the fp64 CPU autograd oracle, over the shapes of
tests/test_gpu_training.py::test_adaptedconv_gradients_vs_autograd_oracle -> a table for DESIGN.md"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_torch as OT
from bridged_gnn_amd import ops, synth
from bridged_gnn_amd.ktgnn import AdaptedConv
DEV = "cuda:0"
rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
worst = {}
for din, D, n in [(24, 16, 300), (20, 7, 257), (48, 64, 400), (32, 128, 500), (16, 2, 1000), (40, 100, 333), (12, 36, 900), (8, 4, 2049), (8, 3, 500), (8, 1, 300),
                  (128, 128, 6000), (128, 2, 6000)]:
    ei, mask = synth.random_multigraph(n, 8 * n, frac_src=0.45, n_isolated=3, seed=n + D)
    rng = np.random.default_rng(D)
    x = rng.standard_normal((n, din)).astype(np.float32); w = rng.standard_normal((n, D)).astype(np.float32)
    torch.manual_seed(D)
    conv = AdaptedConv(din, D, root_weight=False).to(DEV)
    xg = torch.from_numpy(x).to(DEV).requires_grad_(True)
    csr = ops.build_dst_csr(torch.from_numpy(ei).to(DEV), n)
    out = conv(xg, None, central_mask=torch.from_numpy(mask).to(DEV), csr=csr)
    (out * torch.from_numpy(w).to(DEV)).sum().backward()
    p = {k: v.detach().cpu().double().requires_grad_(True) for k, v in conv.state_dict().items()}
    xo = torch.from_numpy(x).double().requires_grad_(True); mo = torch.from_numpy(mask)
    e1, e2 = OT.graph_partition(torch.from_numpy(ei), mo)
    oo = OT.adaptedconv(xo, mo, e1, e2, p)
    (oo * torch.from_numpy(w).double()).sum().backward()
    row = {"out": rel(out.detach().cpu().double(), oo.detach()), "x": rel(xg.grad.cpu().double(), xo.grad)}
    for name, prm in conv.named_parameters():
        row[name] = rel(prm.grad.cpu().double(), p[name].grad)
    print(f"din={din} D={D} n={n}: " + "  ".join(f"{k}={v:.1e}" for k, v in row.items()), flush=True)
    for k, v in row.items(): worst[k] = max(worst.get(k, 0.0), v)
print("WORST: " + "  ".join(f"{k}={v:.1e}" for k, v in worst.items()))
