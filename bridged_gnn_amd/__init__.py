"""MI355X-native (gfx950) implementation of Bridged-GNN's sparse message-passing hot path.

Half A: bridged-graph kNN construction (`bridge.py`) -- reference `main_bridged_graph.py:33-120`.
Half B: KT-GNN `AdaptedConv` aggregation (`ktgnn.py`) -- reference `models/KTGNN.py:218-435`.
Compute goes through the C-ABI library `csrc/libbgnn_hip.so` (declared in `include/bgnn.h`);
there is no CPU fallback: ops raise if the library or a GPU tensor is missing.
"""
from .data import Data, load_bridged_graph, save_bridged_graph  # noqa: F401

__version__ = "0.1.0"
