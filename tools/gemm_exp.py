#!/usr/bin/env python3
"""Timing experiments on transform_gemm_kernel (GPU box; tuning aid). GEMM_EXP=1: no staging after the first tile
(compute + epilogue only), 2: no epilogue stores."""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bridged_gnn_amd import _lib, ops
src = os.path.join(ROOT, "bridged_gnn_amd", "csrc")
def build(exp):
    out = os.path.join(ROOT, "tools", f"libgemm_exp{exp}.so")
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-fno-fast-math", "-Wno-unused-function"]
    if exp: flags.append(f"-DGEMM_EXP={exp}")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + [os.path.join(src, "bgnn_transform.hip"), os.path.join(src, "bgnn_api.hip"), "-o", out])
    return out
dev = "cuda:0"
N, Din, D = 1_000_000, 128, 128
x = torch.randn(N, Din, device=dev)
mask = (torch.arange(N, device=dev) < N // 2).to(torch.uint8)
delta = torch.randn(Din, device=dev)
heads = [{"W_s": torch.randn(D, Din, device=dev) * 0.1, "W_t": torch.randn(D, Din, device=dev) * 0.1, "b_s": torch.randn(D, device=dev),
          "b_t": torch.randn(D, device=dev), "g_s2t": torch.randn(2 * Din, device=dev) * 0.1, "g_t2s": torch.randn(2 * Din, device=dev) * 0.1}]
Wp, bp, gates, D_, ldh, gconst = ops.pack_transform_heads(heads, Din)
o1, o2 = torch.empty(N, ldh, device=dev), torch.empty(N, ldh, device=dev)
small = torch.empty(1024, device=dev)
for exp in [int(a) for a in sys.argv[1:]] or [0, 1]:
    print("NW", os.environ.get("BGNN_GEMM_NW"), end=" ")
    lib = C.CDLL(build(exp))
    fn = lib.bgnn_adaptedconv_transform_f32
    fn.restype, fn.argtypes = _lib.SIGNATURES["bgnn_adaptedconv_transform_f32"]
    def call():
        rc = fn(x.data_ptr(), N, Din, Din, mask.data_ptr(), delta.data_ptr(), 1, D, Wp.data_ptr(), bp.data_ptr(), gates.data_ptr(),
                gconst.data_ptr(), o1.data_ptr(), o2.data_ptr(), None, None, ldh, ldh, small.data_ptr(), None)
        assert rc == 0, rc
    for _ in range(20): call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(40):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); call(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    print("GEMM_EXP", exp, "ms med", round(float(np.median(ts)), 3), "min", round(min(ts), 3), flush=True)
    if exp == 20:           # cycle stamps of transform_wreg_kernel's tile loop, summed over waves and the 60 calls above
        buf = (C.c_ulonglong * 8)()
        torch.cuda.synchronize()
        assert lib.bgnn_debug_wreg_counters(buf) == 0
        tiles = max(int(buf[5]), 1)
        names = ["barrier wait", "stage (sstore + gload issue)", "MFMA chain", "vmcnt(0) wait", "epilogue + stores"]
        print("per wave and tile, s_memtime ticks:", {n: round(int(buf[i]) / tiles, 1) for i, n in enumerate(names)}, "tiles", tiles, flush=True)
        # the same stamps for MODE 2 (bgnn_linear_narrow_transform_f32: 64-row tiles, 4 column waves x 2 row sub-tiles)
        f2 = lib.bgnn_linear_narrow_transform_f32
        f2.restype, f2.argtypes = _lib.SIGNATURES["bgnn_linear_narrow_transform_f32"]
        W0 = torch.randn(128, 128, device=dev) * 0.1; b0 = torch.randn(128, device=dev)
        hd = {"W_s": torch.randn(2, 128, device=dev) * 0.1, "W_t": torch.randn(2, 128, device=dev) * 0.1, "b_s": torch.randn(2, device=dev),
              "b_t": torch.randn(2, device=dev), "g_s2t": torch.randn(256, device=dev) * 0.1, "g_t2s": torch.randn(256, device=dev) * 0.1}
        Wp2, bp2, g2, _, _, _ = ops.pack_transform_heads([hd], 128)
        raw = torch.empty(N, 12, device=dev); cs = torch.zeros(258, dtype=torch.float64, device=dev)
        def call2():
            rc = f2(x.data_ptr(), N, 128, 128, W0.data_ptr(), b0.data_ptr(), 128, 1, mask.data_ptr(), cs.data_ptr(), Wp2.data_ptr(), g2.data_ptr(), raw.data_ptr(), None)
            assert rc == 0, rc
        for _ in range(5): call2()
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record(); call2(); e_.record(); torch.cuda.synchronize(); ts.append(s_.elapsed_time(e_))
        assert lib.bgnn_debug_wreg_counters(buf) == 0
        tiles = max(int(buf[5]), 1)
        print("MODE 2 ms med", round(float(np.median(ts)), 3), "per wave and tile:", {n: round(int(buf[i]) / tiles, 1) for i, n in enumerate(names)}, "tiles", tiles, flush=True)
