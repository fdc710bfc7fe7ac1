"""CPU: pins the torch/autograd oracle (oracle/oracle_torch.py, the gradient checker) -- its forward must equal
the golden vectors produced by the reference's own AdaptedConv."""
import numpy as np
import pytest
import torch

from conftest import assert_close, sub
from oracle import oracle_torch as OT


@pytest.mark.parametrize("D", [2, 31, 64])
def test_torch_oracle_forward_matches_reference_golden(golden, D):
    c = golden(f"conv_small_D{D}.npz")
    mask = torch.from_numpy(c["central_mask"])
    e1, e2 = OT.graph_partition(torch.from_numpy(c["edge_index"].astype(np.int64)), mask)
    p = {k: torch.from_numpy(v) for k, v in sub(c, "p.").items()}
    out = OT.adaptedconv(torch.from_numpy(c["x"]), mask, e1, e2, p)
    assert_close(out.numpy(), c["out"], what=f"torch oracle D={D}")


def test_train_loss_formula():
    torch.manual_seed(0)
    lp = [torch.log_softmax(torch.randn(6, 3), 1) for _ in range(3)]
    y = torch.tensor([0, 1, 2, 0, 1, 2])
    tm = torch.tensor([1, 1, 0, 1, 1, 0], dtype=torch.bool)
    cm = torch.tensor([1, 1, 1, 0, 0, 0], dtype=torch.bool)
    loss = OT.train_loss(lp[0], lp[1], lp[2], y, tm, cm)
    kl = (lp[1].exp() * (lp[1] - lp[2])).sum() / 6
    ref = (2 * (-lp[0][tm, y[tm]]).mean() + (-lp[1][[3, 4], y[[3, 4]]]).mean() + (-lp[2][[3, 4], y[[3, 4]]]).mean()) / 4 + kl
    assert torch.allclose(loss, ref, atol=1e-6)
