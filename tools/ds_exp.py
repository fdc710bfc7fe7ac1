"""domain_sums_kernel: rows per block (BGNN_DS_ROWS) x grid cap (BGNN_DS_GRID) at the full and the rank-sized (1/8) input."""
import os, sys, torch, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from bridged_gnn_amd import ops
    for N in (1_000_000, 125_000):
        D = 128
        x = torch.randn(N, D, device="cuda"); m = (torch.arange(N, device="cuda") < N // 2).to(torch.uint8)
        for _ in range(3): ops.domain_sums(x, m)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): ops.domain_sums(x, m)
        e.record(); torch.cuda.synchronize()
        print("rows", os.environ.get("BGNN_DS_ROWS"), "grid", os.environ.get("BGNN_DS_GRID"), "N", N, "us", 1e3 * s.elapsed_time(e) / 50, flush=True)
else:
    for r in (256, 512, 1024, 2048):
        for g in (256, 512):
            subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, BGNN_DS_GRID=str(g), BGNN_DS_ROWS=str(r)))
