"""`Data` attribute bag + bridged-graph file I/O (host-side mirror of the reference's data surface).

Mirrors the subset of `torch_geometric.data.Data` the reference's two drivers use
(SURVEY.md 8(a16)): construction `main_bridged_graph.py:192-193`, consumption
`main_graph_knowledge_transfer.py:401-411`, `models/KTGNN.py:407-411`.

File format: the reference writes `torch.save(Data)` (`main_bridged_graph.py:317-320`), a torch
zip-pickle whose globals are `torch_geometric.data.data.Data` and
`torch_geometric.data.storage.GlobalStorage` with tensors under `_store._mapping`.  We read it with
`torch.load(weights_only=True)` plus two INERT allow-listed stand-in classes (nothing from the file
is executed, no torch_geometric needed) and write the same layout so PyG users can load our output.
"""
import copy

import torch

__all__ = ["Data", "load_bridged_graph", "save_bridged_graph"]


class Data:
    """Attribute bag: x, edge_index, y, train_mask, val_mask, test_mask, central_mask, ..."""

    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, **kwargs):
        for k, v in dict(x=x, edge_index=edge_index, edge_attr=edge_attr, y=y, **kwargs).items():
            if v is not None:
                setattr(self, k, v)

    # -- introspection ----------------------------------------------------------------------
    @property
    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    @property
    def num_nodes(self):
        if hasattr(self, "x"):
            return int(self.x.shape[0])
        return int(self.edge_index.max()) + 1

    @property
    def num_features(self):
        return int(self.x.shape[1])

    @property
    def num_edges(self):
        return int(self.edge_index.shape[1])

    def __call__(self, *keys):
        """`for name, mask in data('train_mask','val_mask','test_mask')`
        (main_graph_knowledge_transfer.py:80)."""
        for k in (keys or self.keys):
            if hasattr(self, k):
                yield k, getattr(self, k)

    def __repr__(self):
        parts = []
        for k in self.keys:
            v = getattr(self, k)
            parts.append(f"{k}={list(v.shape)}" if torch.is_tensor(v) else f"{k}={v!r}")
        return "Data(" + ", ".join(parts) + ")"

    # -- movement ---------------------------------------------------------------------------
    def to(self, device, non_blocking=False):
        for k in self.keys:
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(self, k, v.to(device, non_blocking=non_blocking))
        return self

    def cpu(self):
        return self.to("cpu")

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def clone(self):
        return copy.deepcopy(self)

    # -- graph ops --------------------------------------------------------------------------
    def coalesce(self):
        """`Data.coalesce()` (main_bridged_graph.py:193): sort by (row, col), drop duplicates."""
        from .utils import coalesce
        self.edge_index = coalesce(self.edge_index, num_nodes=self.num_nodes)
        return self

    def to_undirected_(self):
        """In-place `ToUndirected(merge=True)(data)` -- the reference discards the transform's
        return value (main_graph_knowledge_transfer.py:410-411), i.e. relies on in-place effect."""
        from .utils import to_undirected
        self.edge_index = to_undirected(self.edge_index, num_nodes=self.num_nodes)
        return self


# ---- inert stand-ins for the two PyG classes named inside reference .dat pickles ---------------
class _PygData:
    pass


class _PygGlobalStorage:
    pass


_PygData.__module__, _PygData.__qualname__, _PygData.__name__ = "torch_geometric.data.data", "Data", "Data"
_PygGlobalStorage.__module__ = "torch_geometric.data.storage"
_PygGlobalStorage.__qualname__ = _PygGlobalStorage.__name__ = "GlobalStorage"


def load_bridged_graph(path, map_location="cpu"):
    """Load a reference-format `<name>_bridged_graph.dat` (or one written by
    `save_bridged_graph`) without torch_geometric.  Uses the restricted unpickler
    (`weights_only=True`); `y` is stored tagged `cuda:0` by the reference
    (main_bridged_graph.py:191 vs :167), hence `map_location`."""
    with torch.serialization.safe_globals([_PygData, _PygGlobalStorage]):
        obj = torch.load(path, map_location=map_location, weights_only=True)
    if isinstance(obj, dict):
        mapping = obj
    else:
        mapping = obj.__dict__["_store"].__dict__["_mapping"]
    return Data(**{k: v for k, v in mapping.items()})


def save_bridged_graph(data, path):
    """Write `data` in the reference's on-disk layout (Data -> _store -> _mapping) so that both
    `load_bridged_graph` and a PyG-2.x `torch.load` can read it (main_bridged_graph.py:317-320).
    pickle refers to classes by module path, so while writing (only) the two stand-ins are made
    resolvable under PyG's module names unless the real torch_geometric is importable."""
    import sys
    import types
    store = _PygGlobalStorage()
    d = _PygData()
    store.__dict__["_mapping"] = {k: getattr(data, k).detach().cpu() if torch.is_tensor(getattr(data, k))
                                  else getattr(data, k) for k in data.keys}
    store.__dict__["_parent"] = None
    d.__dict__["_store"] = store
    injected = []
    try:
        for modname, cls in (("torch_geometric", None), ("torch_geometric.data", None),
                             ("torch_geometric.data.data", _PygData), ("torch_geometric.data.storage", _PygGlobalStorage)):
            if modname not in sys.modules:
                sys.modules[modname] = types.ModuleType(modname)
                injected.append(modname)
            if cls is not None and not hasattr(sys.modules[modname], cls.__name__):
                setattr(sys.modules[modname], cls.__name__, cls)
        torch.save(d, path)
    finally:
        for modname in injected:
            sys.modules.pop(modname, None)
