#!/usr/bin/env python3
"""Timing experiments on the kNN pass-1 kernel (GPU box; tuning aid): builds variants of libbgnn with
-DKNN_EXP=n and times bgnn_cosine_topk_f32 on C5.  EXP 1 = no shortlist update, 2 = one barrier per tile
(racy, timing only), 3 = both."""
import ctypes as C, os, subprocess, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bridged_gnn_amd import synth, _lib
src = os.path.join(ROOT, "bridged_gnn_amd", "csrc")
def build(exp):
    out = os.path.join(ROOT, "tools", f"libknn_exp{exp}.so")
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-fno-fast-math", "-Wno-unused-function"]
    if exp: flags.append(f"-DKNN_EXP={exp}")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + [os.path.join(src, "bgnn_knn.hip"), os.path.join(src, "bgnn_api.hip"), "-o", out])
    return out
n = 100_000
dev = "cuda:0"
q = torch.from_numpy(synth.gaussian_embeddings(n, 128, seed=0)).to(dev)
c = torch.from_numpy(synth.gaussian_embeddings(n, 128, seed=1)).to(dev)
qn = q / q.norm(dim=1, keepdim=True); cn = c / c.norm(dim=1, keepdim=True)
for exp in (0, 4):
    lib = C.CDLL(build(exp))
    for name in ("bgnn_topk_workspace_bytes", "bgnn_cosine_topk_f32"):
        fn = getattr(lib, name); fn.restype, fn.argtypes = _lib.SIGNATURES[name]
    wsb = lib.bgnn_topk_workspace_bytes(n, n, 20)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    idx = torch.empty(n, 20, dtype=torch.int64, device=dev); val = torch.empty(n, 20, device=dev)
    nfb = torch.zeros(1, dtype=torch.int32, device=dev)
    ts = []
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = lib.bgnn_cosine_topk_f32(qn.data_ptr(), cn.data_ptr(), n, n, 128, 20, 1, idx.data_ptr(), val.data_ptr(), nfb.data_ptr(), ws.data_ptr(), wsb, None)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("EXP", exp, "rc", rc, "ms", [round(t * 1e3, 2) for t in ts], flush=True)
