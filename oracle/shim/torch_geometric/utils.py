"""Restated PyG utils (oracle shim).  Call sites in the reference: KTGNN.py:299,390-394;
main_bridged_graph.py:75,113,193; utils.py:8."""
import torch


def maybe_num_nodes(edge_index, num_nodes=None):
    if num_nodes is not None:
        return num_nodes
    return int(edge_index.max()) + 1 if edge_index.numel() > 0 else 0


def remove_self_loops(edge_index, edge_attr=None):
    mask = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, mask]
    if edge_attr is None:
        return edge_index, None
    return edge_index, edge_attr[mask]


def add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    N = maybe_num_nodes(edge_index, num_nodes)
    loop = torch.arange(0, N, dtype=torch.long, device=edge_index.device)
    loop = loop.unsqueeze(0).repeat(2, 1)
    edge_index = torch.cat([edge_index, loop], dim=1)
    return edge_index, edge_attr  # edge_attr is None at every reference call site


def coalesce(edge_index, edge_attr=None, num_nodes=None, reduce="add"):
    n = maybe_num_nodes(edge_index, num_nodes)
    key = edge_index[0] * n + edge_index[1]
    key, perm = torch.sort(key, stable=True)
    keep = torch.ones_like(key, dtype=torch.bool)
    keep[1:] = key[1:] != key[:-1]
    out = edge_index[:, perm][:, keep]
    if edge_attr is None:
        return out
    return out, edge_attr


def to_undirected(edge_index, num_nodes=None):
    ei = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    return coalesce(ei, num_nodes=num_nodes)


def degree(index, num_nodes=None, dtype=None):
    N = maybe_num_nodes(index, num_nodes)
    out = torch.zeros((N,), dtype=dtype or torch.float, device=index.device)
    return out.scatter_add_(0, index, torch.ones_like(index, dtype=out.dtype))


def softmax(src, index=None, ptr=None, num_nodes=None, dim=0):
    """scatter-max, exp, scatter-add, divide (+1e-16) -- PyG utils.softmax."""
    N = maybe_num_nodes(index, num_nodes)
    shape = (N,) + tuple(src.shape[1:])
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    src_max = torch.full(shape, float("-inf"), dtype=src.dtype, device=src.device)
    src_max = src_max.scatter_reduce(0, idx, src, reduce="amax", include_self=True)
    out = (src - src_max.gather(0, idx)).exp()
    out_sum = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(0, idx, out)
    return out / (out_sum.gather(0, idx) + 1e-16)
