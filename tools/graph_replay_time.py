import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
from bridged_gnn_amd import synth
from bridged_gnn_amd.data import Data
from bridged_gnn_amd.ktgnn import KTGNN_no_complement
dev = "cuda:0"
for n, e, hid in ((10_000, 140_000, 64), (100_000, 2_000_000, 128), (1_000_000, 20_000_000, 128)):
    ns = n // 2
    extra = max(e - 6 * n - 20 * (n - ns), 0)
    ei, mask = synth.bridged_graph(ns, n - ns, 6, 20, extra, seed=0)
    torch.manual_seed(0)
    model = KTGNN_no_complement(hid, 2, 2, hid, use_bn=True, dim_share=hid).to(dev).eval()
    data = Data(x=torch.randn(n, hid, device=dev), edge_index=torch.from_numpy(ei).to(dev), central_mask=torch.from_numpy(mask).to(dev))
    with torch.no_grad():
        for _ in range(5): model(data)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): model(data)
        torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 50
    run = model.graphed(data)
    for _ in range(5): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): run()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 50
    print(f"N={n} E={e} hidden={hid}: eager {te*1e3:.3f} ms, graph replay {tg*1e3:.3f} ms", flush=True)
