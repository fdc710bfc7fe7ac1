"""Restated attribute-bag `Data` (oracle shim)."""
import copy
import torch
from .utils import coalesce as _coalesce


class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, **kwargs):
        for k, v in dict(x=x, edge_index=edge_index, edge_attr=edge_attr, y=y, **kwargs).items():
            if v is not None:
                setattr(self, k, v)

    @property
    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    @property
    def num_nodes(self):
        return self.x.shape[0]

    @property
    def num_features(self):
        return self.x.shape[1]

    @property
    def num_edges(self):
        return self.edge_index.shape[1]

    def to(self, device):
        for k in self.keys:
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self

    def cpu(self):
        return self.to("cpu")

    def clone(self):
        return copy.deepcopy(self)

    def coalesce(self):
        self.edge_index = _coalesce(self.edge_index, num_nodes=self.num_nodes)
        return self

    def __call__(self, *keys):
        for k in (keys or self.keys):
            if hasattr(self, k):
                yield k, getattr(self, k)

    def __repr__(self):
        return "Data(" + ", ".join(f"{k}={list(getattr(self,k).shape)}" for k in self.keys) + ")"


class InMemoryDataset:  # placeholder (datasets are OUT OF SCOPE)
    pass


def download_url(*a, **k):
    raise RuntimeError("no network")
