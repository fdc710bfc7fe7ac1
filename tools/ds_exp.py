import os, sys, torch, subprocess
sys.path.insert(0, "/root/repo")
if len(sys.argv) > 1:
    from bridged_gnn_amd import ops
    N, D = 1_000_000, 128
    x = torch.randn(N, D, device="cuda"); m = (torch.arange(N, device="cuda") < N // 2).to(torch.uint8)
    for _ in range(3): ops.domain_sums(x, m)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.domain_sums(x, m)
    e.record(); torch.cuda.synchronize()
    print("grid", os.environ.get("BGNN_DS_GRID"), "ms", s.elapsed_time(e) / 20, flush=True)
else:
    for g in (128, 256, 512, 1024):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, BGNN_DS_GRID=str(g)))
