"""edge_index helpers with the torch_geometric names the reference imports
(`remove_self_loops`, `add_self_loops`, `coalesce`, `to_undirected`; reference call sites
models/KTGNN.py:390-394, main_bridged_graph.py:75,:113,:193, main_graph_knowledge_transfer.py:410-411).
GPU tensors only (the sort-based ones call the HIP library)."""
import torch

from . import ops

__all__ = ["remove_self_loops", "add_self_loops", "coalesce", "to_undirected", "set_random_seed", "dataset_conversion",
           "eval_bridged_Graph", "eval_homophily"]


def __getattr__(name):
    # the reference keeps these three in utils.py (:41-131); they live with the graph-assembly code in bridge.py
    if name in ("dataset_conversion", "eval_bridged_Graph", "eval_homophily"):
        from . import bridge
        return getattr(bridge, name)
    raise AttributeError(name)


def _need_cuda(t):
    if not t.is_cuda:
        raise RuntimeError("bridged_gnn_amd.utils works on CUDA(HIP) tensors only; there is no CPU path")


def remove_self_loops(edge_index, edge_attr=None):
    _need_cuda(edge_index)
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep], (None if edge_attr is None else edge_attr[keep])


def add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    _need_cuda(edge_index)
    n = int(num_nodes) if num_nodes is not None else int(edge_index.max().item()) + 1
    loop = torch.arange(n, dtype=torch.int64, device=edge_index.device)
    return torch.cat([edge_index, torch.stack([loop, loop])], dim=1), edge_attr


def coalesce(edge_index, edge_attr=None, num_nodes=None):
    _need_cuda(edge_index)
    if edge_attr is not None:
        raise NotImplementedError("edge_attr is never passed on the reference's hot path")
    return ops.coalesce(edge_index, num_nodes)


def to_undirected(edge_index, num_nodes=None):
    _need_cuda(edge_index)
    return ops.coalesce(torch.cat([edge_index, edge_index.flip(0)], dim=1), num_nodes)


def set_random_seed(seed):
    """reference utils.py:10-17"""
    import random

    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
