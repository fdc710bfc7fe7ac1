"""partitioned trainer at world size 1 (no process group) against the plain single-GPU training step -- quick debugging aid"""
import copy, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from bridged_gnn_amd import synth
from bridged_gnn_amd.data import Data
from bridged_gnn_amd.dist_train import PartitionedTrainer
from bridged_gnn_amd.ktgnn import KTGNN_no_complement
from test_gpu_dist import _ref_loss, _t, DEV
n = 6000
ei, mask = synth.bridged_graph(3500, 2500, 4, 8, 7000, cluster=128, p_local=0.8, seed=4)
for layers in (2, 3):
    torch.manual_seed(0)
    model = KTGNN_no_complement(64, 3, layers, 64, use_bn=True, dim_share=64, dropout=0.0).to(DEV).train()
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(n, 64, device=DEV, generator=g); y = torch.randint(0, 3, (n,), device=DEV, generator=g)
    tm = torch.rand(n, device=DEV, generator=g) < 0.5; cm = _t(mask)
    data = Data(x=x, edge_index=_t(ei), central_mask=cm)
    ref = copy.deepcopy(model)
    tr = PartitionedTrainer(model, ei, mask, 0, 1, DEV)
    own = tr.owned_global
    out_r = ref(data); loss_r = _ref_loss(out_r, y, tm, cm, n); loss_r.backward()
    out_p = tr.forward(x[own].contiguous()); loss_p = tr.reference_loss(out_p, y[own], tm[own]); loss_p.backward(); tr.sync_grads()
    print("layers", layers, "loss", float(loss_r), float(loss_p), "out", max(float((a - b[own]).abs().max()) for a, b in zip(out_p, out_r[:3])))
    for (nm, p), r in zip(model.named_parameters(), ref.parameters()):
        e = float((p.grad - r.grad).abs().max()) / (float(r.grad.abs().max()) + 1e-6)
        if e > 1e-3: print("   grad mismatch", nm, e)
print("done")
