"""The reference's step-2 schedule on the config-3 stand-in (300 epochs, Adam lr 1e-3 wd 5e-3, dropout 0.5, eval every 10 epochs;
main_graph_knowledge_transfer.py:143-262): eager loop vs one HIP graph per training step, same seeds.  Prints wall time and the
loss / accuracy curves' end points (the two runs see different dropout masks, so they agree statistically, not bit for bit)."""
import os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bridged_gnn_amd import synth, utils
from bridged_gnn_amd.data import Data
from bridged_gnn_amd.ktgnn import KTGNN_no_complement
dev = "cuda:0"
x, ei, y, m = synth.twitter_standin(seed=0)
n = x.shape[0]
und = utils.to_undirected(torch.from_numpy(ei).to(dev), n)
data = Data(x=torch.from_numpy(x).to(dev), edge_index=und, central_mask=torch.from_numpy(m).to(dev))
yt = torch.from_numpy(y).to(dev)
lab = yt >= 0
g = torch.Generator(device=dev).manual_seed(0)
tr = (torch.rand(n, device=dev, generator=g) < 0.6) & lab
te = ~tr & lab
cm = data.central_mask
yi = yt.clamp_min(0)[:, None]
w_b = tr.float() / tr.sum(); w_t = (tr & ~cm).float() / (tr & ~cm).sum().clamp_min(1)
nll = lambda lp, w: -(lp.gather(1, yi).squeeze(1) * w).sum()


def loss_fn(o):
    lb, lt, lth, _ = o
    return (2 * nll(lb, w_b) + nll(lt, w_t) + nll(lth, w_t)) / 4 + F.kl_div(lth, lt, log_target=True, reduction="batchmean")


def evaluate(model):
    model.eval()
    with torch.no_grad():
        lb, lt, lth, _ = model(data)
    model.train()
    pred = torch.where(cm, lb.argmax(1), lt.argmax(1))
    return float((pred[te] == yt[te]).float().mean())


for mode in ("eager", "graphed"):
    torch.manual_seed(0)
    model = KTGNN_no_complement(300, 2, 2, 128, use_bn=True, dim_share=300, dropout=0.5).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-3, capturable=True)
    if mode == "graphed":
        t0 = time.perf_counter(); step = model.graphed_train_step(data, loss_fn, opt); torch.cuda.synchronize(); tcap = time.perf_counter() - t0
    else:
        tcap = 0.0

        def step():
            opt.zero_grad(set_to_none=True); l = loss_fn(model(data)); l.backward(); opt.step(); return l
    torch.cuda.synchronize(); t0 = time.perf_counter()
    accs, losses = [], []
    for ep in range(300):
        l = step()
        if ep % 10 == 9:
            losses.append(float(l.detach())); accs.append(evaluate(model))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ok = all(v == v for v in losses)
    print(f"{mode:8s} 300 epochs + 30 evals: {t1 - t0:.3f} s (+ {tcap:.2f} s warm-up/capture) | loss {losses[0]:.4f} -> {losses[-1]:.4f} | test acc {accs[0]:.3f} -> {accs[-1]:.3f} | finite {ok}", flush=True)
