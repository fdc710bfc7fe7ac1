"""The wide aggregation inside a captured HIP graph whose inputs change per replay, with eager work between replays.
Variants on the command line: plain | fn | fn+grad | plain+side | fn+grad+side (fn: through the autograd Function, grad: leaves require
grad, side: warm-up on a side stream).  With hipMemsetAsync clearing the tile counters, every `+side` variant froze from replay 1 on
(the memset node replayed a stale pattern); with bgnn_zero_async all variants are exact.  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridged_gnn_amd import ops, synth, utils
from bridged_gnn_amd.ktgnn import _AggregateFn
dev = "cuda:0"
x_np, ei, y, m = synth.twitter_standin(seed=0)
n = x_np.shape[0]
und = utils.to_undirected(torch.from_numpy(ei).to(dev), n)
m8 = torch.from_numpy(m).to(dev).to(torch.uint8)
csr = ops.build_dst_csr(und, n)
g = torch.Generator(device=dev).manual_seed(1)
D = 128
rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-20))
def run(variant):
    h1, h2 = torch.randn(n, D, device=dev, generator=g), torch.randn(n, D, device=dev, generator=g)
    a1, a2 = torch.randn(D, device=dev, generator=g) * 0.1, torch.randn(D, device=dev, generator=g) * 0.1
    if "grad" in variant:
        for t in (h1, h2, a1, a2): t.requires_grad_(True)
    if "fn" in variant:
        f = lambda A, B, C, E: _AggregateFn.apply(A, B, C, E, csr, m8, D, 0.1)
    else:
        f = lambda A, B, C, E: ops.adaptedconv_aggregate(A, B, C, E, csr, m8, D, 0.1, want_alpha=True)[0]
    st = torch.cuda.Stream() if "side" in variant else torch.cuda.current_stream()
    if "side" in variant: st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for _ in range(2): f(h1, h2, a1, a2)
    torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        with torch.no_grad():
            h1.mul_(0.97); h2.mul_(0.97)
        res = f(h1, h2, a1, a2)
    errs = []
    for i in range(4):
        gr.replay(); torch.cuda.synchronize()
        ref = f(h1.detach().clone(), h2.detach().clone(), a1.detach().clone(), a2.detach().clone())
        errs.append(f"{rel(res.detach(), ref.detach()):.1e}")
    print(variant.ljust(18), errs, flush=True)
for v in sys.argv[1:]:
    run(v)
