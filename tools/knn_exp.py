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
    if isinstance(exp, tuple):
        flags += [f"-D{k}={v}" for k, v in exp]
        out = os.path.join(ROOT, "tools", "libknn_exp_" + "_".join(f"{k}{v}" for k, v in exp) + ".so")
    elif exp: flags.append(f"-DKNN_EXP={exp}")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + [os.path.join(src, "bgnn_knn.hip"), os.path.join(src, "bgnn_api.hip"), "-o", out])
    return out
n = 100_000
dev = "cuda:0"
q = torch.from_numpy(synth.gaussian_embeddings(n, 128, seed=0)).to(dev)
c = torch.from_numpy(synth.gaussian_embeddings(n, 128, seed=1)).to(dev)
qn = q / q.norm(dim=1, keepdim=True); cn = c / c.norm(dim=1, keepdim=True)
def _parse(a):
    return int(a) if a.isdigit() else tuple((kv.split('=')[0], kv.split('=')[1]) for kv in a.split(','))
for exp in ([_parse(a) for a in sys.argv[1:]] or [0, 1]):
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
    if exp in (8, 9, 10):
        buf = (C.c_ulonglong * 8)()
        lib.bgnn_debug_knn_counters.argtypes = [C.c_void_p]
        print("counters rc", lib.bgnn_debug_knn_counters(buf), "slow_tiles rounds queued drains compactions:", [int(x) / 4 for x in buf][:8], "(per call)"); print("clock GHz ~", buf[6] / max(buf[7], 1) * 0.1)
        bb = (C.c_ulonglong * 3072)()
        lib.bgnn_debug_knn_blocks.argtypes = [C.c_void_p]
        lib.bgnn_debug_knn_blocks(bb)
        a = np.array(list(bb), dtype=np.uint64).reshape(1024, 3)[:512]
        st, en = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
        print('first blocks', ((st - st.min())[:4] / 100.0), ((en - st.min())[:4] / 100.0), 'blocks 256..259', ((st - st.min())[256:260] / 100.0), ((en - st.min())[256:260] / 100.0))
        t0 = st.min()
        print("block start (us) pctl", np.percentile((st - t0) / 100.0, [0, 25, 50, 75, 100]))
        print("block end   (us) pctl", np.percentile((en - t0) / 100.0, [0, 25, 50, 75, 100]))
        print("late starters (>1ms):", int(((st - t0) > 100000).sum()))
        hw = a[:, 2]
        xcc = (hw >> np.uint64(32)) & np.uint64(0xF); hwid = hw & np.uint64(0xFFFFFFFF)
        cu = (hwid >> np.uint64(8)) & np.uint64(0xF); se = (hwid >> np.uint64(13)) & np.uint64(0x7); sh = (hwid >> np.uint64(12)) & np.uint64(1)
        key = xcc * np.uint64(1000) + se * np.uint64(100) + sh * np.uint64(50) + cu
        import collections
        cnt = collections.Counter(key.tolist())
        print("distinct CU keys", len(cnt), "max blocks on one", max(cnt.values()), "hist", collections.Counter(cnt.values()))
