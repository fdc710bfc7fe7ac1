import sys, time, torch, argparse
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bridged_gnn_amd import ops
args = argparse.Namespace(config="c4", nodes=1_000_000, edges=20_000_000, graph="local", feat=128, hidden=128, classes=2)
wl = bench.make_workload(args, torch.device("cuda:0"))
ei = torch.from_numpy(wl["ei_np"]).cuda()
n = wl["mask_np"].shape[0]
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    csr = ops.build_dst_csr(ei, n)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); csr = ops.build_dst_csr(ei, n); e.record(); torch.cuda.synchronize()
    print(f"call {i}: wall {1e3*(t1-t0):.2f} ms, events {s.elapsed_time(e):.2f} ms, E'={csr.num_edges}", flush=True)
