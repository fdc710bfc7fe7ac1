import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import argparse, torch
import bench
from bridged_gnn_amd.data import Data
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
sys.argv = [sys.argv[0], "--config", cfg]
args = bench.parse()
dev = torch.device("cuda:0")
wl = bench.make_workload(args, dev)
model = bench.build_model(args, dev)
data = Data(x=wl["x"], edge_index=torch.from_numpy(wl["ei_np"]).to(dev), central_mask=torch.from_numpy(wl["mask_np"]).to(dev))
with torch.no_grad():
    for _ in range(20): model(data)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300): model(data)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
