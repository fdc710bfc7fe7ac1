"""host-side enqueue time of one eval forward (the GPU idles whenever this exceeds the GPU time of a forward)"""
import sys, os, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sys.argv = ["bench.py", "--no-cpu", "--no-knn", "--train-steps", "0"]
args = bench.parse()
dev = torch.device("cuda:0")
wl = bench.make_workload(args, dev)
from bridged_gnn_amd.data import Data
model = bench.build_model(args, dev)
data = Data(x=wl["x"], edge_index=torch.from_numpy(wl["ei_np"]).to(dev), central_mask=torch.from_numpy(wl["mask_np"]).to(dev))
with torch.no_grad():
    seq = []
    for _ in range(40):
        torch.cuda.synchronize(); t0 = time.perf_counter(); model(data); torch.cuda.synchronize(); seq.append((time.perf_counter() - t0) * 1e3)
    print("synchronised ms per forward, first 40 of the process:", [round(t, 2) for t in seq])
    print("allocator:", {k: v for k, v in torch.cuda.memory_stats().items() if k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "reserved_bytes.all.current", "allocated_bytes.all.peak")})
    t0 = time.perf_counter()
    for _ in range(20): model(data)
    torch.cuda.synchronize(); print("20 back-to-back forwards: ms each", (time.perf_counter() - t0) * 1e3 / 20)
    print("allocator:", {k: v for k, v in torch.cuda.memory_stats().items() if k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "reserved_bytes.all.current")})
    for _ in range(5): model(data)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); model(data); ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    print("host enqueue ms per forward:", [round(t, 2) for t in ts])
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): model(data)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
