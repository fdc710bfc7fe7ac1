"""Replays of the captured eval forward (model.graphed) on configs 2 and 3 against the eager forward, several replays in a row.  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridged_gnn_amd import synth, utils
from bridged_gnn_amd.data import Data
from bridged_gnn_amd.ktgnn import KTGNN_no_complement
dev = "cuda:0"
for cfg in ("c3", "c2"):
    if cfg == "c3":
        x, ei, y, m = synth.twitter_standin(seed=0); feat, hidden = 300, 128
    else:
        x, ei, y, m = synth.sync_rd_intra(n=10000, feat=128, homophily=0.7, deg=10, k_cross=20, seed=0); feat, hidden = 128, 64
    und = utils.to_undirected(torch.from_numpy(ei).to(dev), x.shape[0])
    data = Data(x=torch.from_numpy(x).to(dev), edge_index=und, central_mask=torch.from_numpy(m).to(dev))
    torch.manual_seed(0)
    model = KTGNN_no_complement(feat, 2, 2, hidden, use_bn=True, dim_share=feat).to(dev).eval()
    with torch.no_grad():
        ref = [t.clone() for t in model(data)[:3]]
        run = model.graphed(data)
        for i in range(4):
            out = run(); torch.cuda.synchronize()
            print(cfg, "replay", i, "max abs diff vs eager", max(float((a - b).abs().max()) for a, b in zip(out[:3], ref)), flush=True)
