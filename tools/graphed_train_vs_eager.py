"""graphed_train_step vs the same loop run eagerly on a twin model, interleaved step by step (adam | adam_single | adam_fused | sgd_mom |
adam_eager_twin = eager vs eager as the control); prints the steps where the losses differ by more than 1e-3 and the last one.  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
DEV = "cuda:0"
import test_gpu_round2 as T
data, loss_fn, a, b = T._small_training_setup(0.0)
kind = sys.argv[1] if len(sys.argv) > 1 else "adam"
mk = {"adam": lambda ps: torch.optim.Adam(ps, lr=1e-3, capturable=True),
      "adam_single": lambda ps: torch.optim.Adam(ps, lr=1e-3, capturable=True, foreach=False),
      "adam_fused": lambda ps: torch.optim.Adam(ps, lr=1e-3, capturable=True, fused=True),
      "sgd_mom": lambda ps: torch.optim.SGD(ps, lr=0.01, momentum=0.9),
      "adam_eager_twin": lambda ps: torch.optim.Adam(ps, lr=1e-3, capturable=True)}[kind]
oa, ob = mk(a.parameters()), mk(b.parameters())
W = 2
def eager(m, o):
    o.zero_grad(set_to_none=True); l = loss_fn(m(data)); l.backward(); o.step(); return float(l.detach())
if kind != "adam_eager_twin":
    run = a.graphed_train_step(data, loss_fn, oa, warmup=W)
    stepa = lambda: float(run().detach())
else:
    for _ in range(W): eager(a, oa)
    stepa = lambda: eager(a, oa)
for _ in range(W): eager(b, ob)
for i in range(60):
    la, lb = stepa(), eager(b, ob)
    pd = sorted(((float((p.detach() - q.detach()).abs().max()), n) for (n, p), q in zip(a.named_parameters(), b.parameters())), reverse=True)[:3]
    gd = sorted(((float((p.grad - q.grad).abs().max() / (q.grad.abs().max() + 1e-20)), n) for (n, p), q in zip(a.named_parameters(), b.parameters())), reverse=True)[:3]
    if abs(la - lb) > 1e-3 * abs(lb) or i == 59: print(kind, i, round(la, 5), round(lb, 5), "| param abs diff", [(f"{d:.1e}", n) for d, n in pd], "| grad rel diff", [(f"{d:.1e}", n) for d, n in gd], flush=True)
