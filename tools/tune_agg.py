#!/usr/bin/env python3
"""Sweep (LF, EP, U) instantiations of agg_kernel on the C4 graph (GPU box only; tuning aid).
Build: hipcc -O3 --offload-arch=gfx950 -fPIC -shared -fno-fast-math tools/tune_agg.hip -o tools/libbgnn_tune.so"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bridged_gnn_amd import ops, synth  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "tools", os.environ.get("TUNE_LIB", "libbgnn_tune.so")))
P, I64, I32, F32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float
lib.bgnn_tune_aggregate.restype = C.c_int
lib.bgnn_tune_aggregate.argtypes = [P, P, I64, P, P, P, P, P, I64, I64, I32, F32, P, I64, C.c_int, P, P]
dev = "cuda:0"


def run(graph, D, variants, n=1_000_000):
    ns = n // 2
    ei, mask = synth.bridged_graph(ns, n - ns, 6, 20, 4_000_000, p_local=0.9 if graph == "local" else 0.0, seed=0)
    print('edges', ei.shape, flush=True)
    csr = ops.build_dst_csr(torch.from_numpy(ei).to(dev), n)
    ld = ops.pad4(D)
    hS = torch.zeros(n, ld, device=dev); hT = torch.zeros(n, ld, device=dev)
    hS[:, :D] = torch.randn(n, D, device=dev); hT[:, :D] = torch.randn(n, D, device=dev)
    a1 = torch.randn(D, device=dev); a2 = torch.randn(D, device=dev)
    m8 = torch.from_numpy(mask).to(dev).to(torch.uint8)
    out = torch.empty(n, ld, device=dev)
    ref = None
    res = {}
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    queue = torch.zeros(8, dtype=torch.int32, device=dev)
    qp = queue.data_ptr() if os.environ.get('TUNE_QUEUE', '1') == '1' else None
    for v in variants:
        def call():
            lib.bgnn_tune_reset_counters(None)
            rc = lib.bgnn_tune_aggregate(hS.data_ptr(), hT.data_ptr(), ld, a1.data_ptr(), a2.data_ptr(), csr.rowptr.data_ptr(),
                                         csr.col.data_ptr(), m8.data_ptr(), 0, n, D, 0.1, out.data_ptr(), ld, v, qp, st)
            assert rc == 0, rc
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); call(); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        if ref is None:
            ref = out.clone()
        ok = bool(torch.allclose(out, ref, rtol=1e-5, atol=1e-6))
        res[v] = {"ms_med": float(np.median(ts)), "ms_min": float(min(ts)), "same": ok}
        print(graph, D, v, res[v], flush=True)
    return res


if __name__ == "__main__":
    allres = {}
    if os.environ.get("TUNE_C4ONLY"):
        allres["local_D128"] = run("local", 128, [40, 41])
        allres["uniform_D128"] = run("uniform", 128, [40, 41])
    elif os.environ.get("TUNE_SMALL"):
        # L2-resident tables: is the kernel or the memory system the bound?
        for n in (4096, 32768, 262144):
            allres[f"local_D128_n{n}"] = run("local", 128, [0, 40, 41], n=n)
            allres[f"uniform_D128_n{n}"] = run("uniform", 128, [40], n=n)
    elif os.environ.get("TUNE_QUICK"):
        allres["local_D128"] = run("local", 128, [0, 40, 41])
        allres["uniform_D128"] = run("uniform", 128, [0, 40, 41])
        allres["local_D64"] = run("local", 64, [23, 42, 43])
        allres["local_D100"] = run("local", 100, [0, 40, 41])
        allres["local_D256"] = run("local", 256, [6, 44, 45], n=500_000)
    else:
        for graph in ("local", "uniform"):
            allres[f"{graph}_D128"] = run(graph, 128, [0, 1, 2, 3, 4, 5])
            allres[f"{graph}_D2"] = run(graph, 2, [10, 11, 12, 13, 14, 15, 16, 17])
        allres["local_D64"] = run("local", 64, [20, 21, 22, 23])
        allres["local_D31"] = run("local", 31, [30, 31, 32, 33])
    json.dump(allres, open(os.path.join(ROOT, "gpurun_out", "tune_agg.json"), "w"), indent=1)
