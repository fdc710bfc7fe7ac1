"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU torch (fp32/fp64, autograd) restatement of AdaptedConv /
KTGNN_no_complement in the reference's op order (index_select / elementwise / index_add_), used as the
checker for GRADIENTS (tests/test_gpu_training.py).  Its forward is pinned against oracle_np / the golden
vectors in tests/test_oracle_torch.py.  Citations relative to /root/reference/Bridged-GNN/."""
import torch
import torch.nn.functional as F


def segment_softmax(src, index, n):
    """torch_geometric.utils.softmax (call site models/KTGNN.py:299)"""
    m = torch.full((n,), float("-inf"), dtype=src.dtype).scatter_reduce(0, index, src, reduce="amax", include_self=True)
    e = (src - m[index]).exp()
    s = torch.zeros(n, dtype=src.dtype).index_add_(0, index, e)
    return e / (s[index] + 1e-16)


def adaptedconv(x, mask, e1, e2, p, slope=0.1):
    """models/KTGNN.py:263-315 (root_weight=False).  p: dict of tensors (may require grad)."""
    n = x.shape[0]
    diff = (x[mask].mean(0, keepdim=True) - x[~mask].mean(0, keepdim=True)).expand(x.shape)
    cat = torch.cat((x, diff), -1)
    s2t = torch.tanh(cat @ p["a_g_s2t.weight"].t()) * diff
    t2s = torch.tanh(cat @ p["a_g_t2s.weight"].t()) * diff
    h_s2t = F.linear(x - s2t * mask.unsqueeze(-1), p["lin_t.weight"], p.get("lin_t.bias"))
    h_t2s = F.linear(x + t2s * (~mask).unsqueeze(-1), p["lin_s.weight"], p.get("lin_s.bias"))
    a1 = F.leaky_relu(h_t2s[e1[0]] + h_t2s[e1[1]], slope) @ p["a_f_t2s.weight"].reshape(-1)
    a2 = F.leaky_relu(h_s2t[e2[0]] + h_s2t[e2[1]], slope) @ p["a_f_s2t.weight"].reshape(-1)
    alpha = segment_softmax(torch.cat((a1, a2)), torch.cat((e1[1], e2[1])), n)
    out = torch.zeros(n, h_s2t.shape[1], dtype=x.dtype)
    out = out.index_add(0, e1[1], h_t2s[e1[0]] * alpha[: e1.shape[1], None])
    out = out.index_add(0, e2[1], h_s2t[e2[0]] * alpha[e1.shape[1]:, None])
    return out


def graph_partition(edge_index, mask):
    n = mask.shape[0]
    ei = edge_index[:, edge_index[0] != edge_index[1]]
    loop = torch.arange(n)
    ei = torch.cat([ei, torch.stack([loop, loop])], 1)
    m1 = mask[ei[1]]
    return ei[:, m1], ei[:, ~m1]


def train_loss(logp_s, logp_t, logp_that, y, train_mask, central_mask, Lambda=1.0):
    """main_graph_knowledge_transfer.py:44-54"""
    tm_t = train_mask & ~central_mask
    l_s = F.nll_loss(logp_s[train_mask], y[train_mask])
    l_t1 = F.nll_loss(logp_t[tm_t], y[tm_t])
    l_t2 = F.nll_loss(logp_that[tm_t], y[tm_t])
    l_kl = F.kl_div(logp_that, logp_t, log_target=True, reduction="batchmean")
    return (l_s * 2.0 + l_t1 + l_t2) / 4.0 + l_kl * Lambda
