"""ORACLE-ONLY shim of the third-party `torch_geometric` package (absent from this image).

Test infrastructure: exists solely so that `oracle/gen_golden.py` can import the reference
(/root/reference/Bridged-GNN/*.py) IN THE BUILD CONTAINER and emit golden vectors into
tests/golden/.  It never travels to the product path and nothing under bridged_gnn_amd/ imports it.
Semantics restated from SURVEY.md Appendix A / C (PyG 2.0-2.3 era); version unpinned by the reference.
"""
from . import typing, utils, data, transforms, nn  # noqa: F401
__version__ = "0.0-oracle-shim"
