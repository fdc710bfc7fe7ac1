"""ORACLE-ONLY stand-in for the absent third-party `torch_sparse` (test infrastructure; never imported by the product).

`SparseTensor` / `matmul` restate the two call shapes the reference's `utils.py:101-131` uses -- a COO matrix of ones built from
`row=`, `col=`, `value=`, `sparse_sizes=`; `matmul(A, dense, reduce='sum')` (the default reduce) and `A.to_dense()` -- following
torch_sparse's published semantics (out[row] += value * dense[col]; duplicate entries add up).  The other names are import
placeholders (`models/KTGNN.py:18-19`, `models/models.py:18` import them, the hot path never calls them).
"""
import torch


class SparseTensor:
    def __init__(self, row=None, col=None, value=None, sparse_sizes=None, **kw):
        if row is None or col is None or sparse_sizes is None:
            raise NotImplementedError("oracle shim: SparseTensor(row=, col=, value=, sparse_sizes=) only")
        self.row, self.col = row.reshape(-1).long(), col.reshape(-1).long()
        self.value = value if value is not None else torch.ones(self.row.shape[0], device=self.row.device)
        self.sizes = tuple(int(s) for s in sparse_sizes)

    def to_dense(self):
        out = torch.zeros(self.sizes, dtype=self.value.dtype, device=self.value.device)
        out.index_put_((self.row, self.col), self.value, accumulate=True)
        return out

    def sparse_sizes(self):
        return self.sizes


def _matmul(src, other, reduce="sum"):
    if reduce not in ("sum", "add"):
        raise NotImplementedError("oracle shim: reduce='sum' only")
    if isinstance(other, SparseTensor):
        other = other.to_dense()
    out = torch.zeros((src.sizes[0],) + tuple(other.shape[1:]), dtype=other.dtype, device=other.device)
    out.index_add_(0, src.row, other[src.col] * src.value.to(other.dtype).reshape(-1, *([1] * (other.dim() - 1))))
    return out


def _ph(*a, **k):
    raise NotImplementedError("oracle shim placeholder")


fill_diag = sum = mul = set_diag = _ph
from . import matmul as _mm_module      # noqa: E402  (`import torch_sparse.matmul as matmul`, utils.py:6, needs the submodule ...)
matmul = _matmul                        # ... while the package attribute is the FUNCTION, as in torch_sparse itself
