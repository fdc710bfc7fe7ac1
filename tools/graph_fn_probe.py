"""Which autograd.Function misbehaves inside a captured HIP graph?  For each piece: capture `inputs *= c; out = f(inputs); (out * w).sum().backward()`
and compare every replay's output / gradients with an eager evaluation on the same (current) inputs."""
import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridged_gnn_amd import ops, synth, utils
from bridged_gnn_amd.ktgnn import (KTGNN_no_complement, _AggregateFn, _AggregateHeadsFn, _BnReluDropFn, _LinearFn, _as_u8)
dev = "cuda:0"
x_np, ei, y, m = synth.twitter_standin(seed=0)
n = x_np.shape[0]
und = utils.to_undirected(torch.from_numpy(ei).to(dev), n)
mask = torch.from_numpy(m).to(dev)
m8 = mask.to(torch.uint8)
torch.manual_seed(0)
model = KTGNN_no_complement(300, 2, 2, 128, use_bn=True, dim_share=300, dropout=0.0).to(dev).train()
from bridged_gnn_amd.data import Data
csr = model._prepare(Data(x=torch.from_numpy(x_np).to(dev), edge_index=und, central_mask=mask))
g = torch.Generator(device=dev).manual_seed(1)
R = lambda *s: torch.randn(*s, device=dev, generator=g)
rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-20))


def probe(name, leaves, fn, nrep=12):
    """leaves: list of tensors (requires_grad set here); fn(leaves) -> output tensor"""
    leaves = [t.clone().requires_grad_(True) for t in leaves]
    w = None
    side = torch.cuda.current_stream() if os.environ.get("NOSIDE") else torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            out = fn(leaves)
            w = torch.randn(out.shape, device=dev, generator=g) if w is None else w
            (out * w).sum().backward()
            for t in leaves: t.grad = None
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        with torch.no_grad():
            for t in leaves: t.mul_(0.97)
        out = fn(leaves)
        if not os.environ.get("NOBWD"): (out * w).sum().backward()
    worst = 0.0
    for i in range(nrep):
        gr.replay(); torch.cuda.synchronize()
        ref_leaves = [t.detach().clone().requires_grad_(True) for t in leaves]
        ro = fn(ref_leaves); (ro * w).sum().backward()
        errs = [rel(out.detach(), ro.detach())] + ([] if os.environ.get('NOBWD') else [rel(t.grad, r.grad) for t, r in zip(leaves, ref_leaves) if r.grad is not None])
        worst = max(worst, max(errs))
        if max(errs) > 1e-3 or any(e != e for e in errs):
            bad = ((out.detach() - ro.detach()).abs().amax(dim=tuple(range(1, out.dim()))) > 1e-3 * ro.detach().abs().max()).nonzero().flatten()
            deg = (csr.rowptr[1:] - csr.rowptr[:-1])
            print("  ", name, "replay", i, "errs", [f"{e:.1e}" for e in errs], "| bad output rows", int(bad.numel()), "of", out.shape[0],
                  "| of them hubs (deg>=128):", int((deg[bad] >= 128).sum()) if bad.numel() and out.shape[0] == n else "-", flush=True)
    print(name, "worst error over", nrep, "replays:", f"{worst:.1e}", flush=True)


D = 128
ONLY = os.environ.get("ONLY")
if ONLY:
    _p = probe
    probe = lambda name, *a, **k: _p(name, *a, **k) if ONLY in name else None
probe("aggregate wide D=128", [R(n, D), R(n, D), R(D) * 0.1, R(D) * 0.1],
      lambda L: _AggregateFn.apply(L[0], L[1], L[2], L[3], csr, m8, D, 0.1))
tabs = []
for _ in range(6):
    t = torch.zeros(n, 4, device=dev); t[:, :2] = R(n, 2); tabs.append(t)
probe("aggregate heads", [R(3, 2) * 0.5, R(3, 2) * 0.5] + tabs,
      lambda L: _AggregateHeadsFn.apply(csr, m8, 2, 0.1, *L)[:, :, :2])
bn = copy.deepcopy(model.bns[0])
probe("bn relu dropout", [R(n, D), bn.weight.detach(), bn.bias.detach()], lambda L: _BnReluDropFn.apply(L[0], L[1], L[2], bn, True, 0.0))
probe("linear", [R(n, D), R(D, D) * 0.1, R(D) * 0.1], lambda L: _LinearFn.apply(L[0], L[1], L[2]))
conv = model.convs[0]
xin = torch.from_numpy(x_np).to(dev)


def tr(L):
    conv.lin_s.weight.data, conv.lin_t.weight.data = L[1].data, L[2].data
    from bridged_gnn_amd.ktgnn import _TransformFn
    a, b = _TransformFn.apply(L[0], L[1], conv.lin_s.bias, L[2], conv.lin_t.bias, L[3], L[4], m8, conv, None)
    return torch.cat((a, b), 1)
probe("transform 300->128", [xin, conv.lin_s.weight.detach(), conv.lin_t.weight.detach(), conv.a_g_s2t.weight.detach(), conv.a_g_t2s.weight.detach()], tr)
c2 = model.clf_base
probe("transform 128->2", [R(n, D), c2.lin_s.weight.detach(), c2.lin_t.weight.detach(), c2.a_g_s2t.weight.detach(), c2.a_g_t2s.weight.detach()],
      lambda L: (lambda ab: torch.cat(ab, 1))(__import__("bridged_gnn_amd.ktgnn", fromlist=["_TransformFn"])._TransformFn.apply(
          L[0], L[1], c2.lin_s.bias, L[2], c2.lin_t.bias, L[3], L[4], m8, c2, None)))
