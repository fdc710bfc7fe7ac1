"""config C3 forward for (hub threshold, segment) pairs -> picks ops.HUB_THRESHOLD / HUB_SEGMENT"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridged_gnn_amd import ops, synth, utils
from bridged_gnn_amd.data import Data
from bridged_gnn_amd.ktgnn import KTGNN_no_complement
dev = "cuda:0"
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
if cfg == "c3":
    x, ei, y, m = synth.twitter_standin(seed=0); feat, hidden = 300, 128
else:
    x, ei, y, m = synth.sync_rd_intra(n=10000, feat=128, homophily=0.7, deg=10, k_cross=20, seed=0); feat, hidden = 128, 64
und = utils.to_undirected(torch.from_numpy(ei).to(dev), x.shape[0])
data = Data(x=torch.from_numpy(x).to(dev), edge_index=und, central_mask=torch.from_numpy(m).to(dev))
for thr, seg in ((192, 64), (128, 64), (1 << 30, 64), (96, 32), (96, 48), (128, 32), (64, 32), (80, 40), (160, 64)):
    ops.HUB_THRESHOLD, ops.HUB_SEGMENT = thr, seg
    ops.DstCSR.hub_tables.__defaults__ = (thr, seg)
    torch.manual_seed(0)
    model = KTGNN_no_complement(feat, 2, 2, hidden, use_bn=True, dim_share=feat).to(dev).eval()
    with torch.no_grad():
        run = model.graphed(data)
        for _ in range(5): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): run()
        torch.cuda.synchronize()
        hubs = model._csr.hub_tables(thr, seg)
    print(f"threshold {thr} segment {seg}: {1e3 * (time.perf_counter() - t0) / 100:.4f} ms per forward (graph replay), hubs {0 if hubs is None else hubs[0].numel()}, segments {0 if hubs is None else hubs[3].numel()}", flush=True)
