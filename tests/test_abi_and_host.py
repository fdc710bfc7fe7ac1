"""CPU: the C-ABI library loads and exports every symbol include/bgnn.h declares (no compute calls
without a GPU); host-side logic (Data, file round trip, refusal of host tensors, synthetic generators)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "bgnn.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bgnn_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from bridged_gnn_amd import _lib
    names = _declared()
    assert len(names) >= 15
    lib = ctypes.CDLL(_lib.SO_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in bgnn.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes SIGNATURES out of sync with include/bgnn.h"
    l = _lib.lib()
    assert l.bgnn_version() == _lib.ABI_VERSION == 113
    assert b"NULL" in l.bgnn_error_string(-1) and l.bgnn_error_string(0) == b"success"


def test_ops_fail_loudly_without_gpu_tensors():
    from bridged_gnn_amd import ops, utils
    from bridged_gnn_amd.ktgnn import AdaptedConv, KTGNN_no_complement
    ei = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises(RuntimeError, match="no CPU"):
        ops.build_dst_csr(ei, 2)
    with pytest.raises(RuntimeError, match="no CPU"):
        utils.to_undirected(ei)
    conv = AdaptedConv(4, 4, root_weight=False)
    with pytest.raises(RuntimeError, match="no CPU"):
        with torch.no_grad():
            conv(torch.zeros(2, 4), ei, ei, ei, torch.tensor([True, False]))
    with pytest.raises(NotImplementedError):
        KTGNN_no_complement(4, need_complement=True)


def test_state_dict_keys_match_reference_layout(golden):
    from bridged_gnn_amd.ktgnn import KTGNN_no_complement
    k = golden("ktgnn_office.npz")
    ref_keys = sorted(n[3:] for n in k if n.startswith("sd."))
    m = KTGNN_no_complement(256, 31, 2, 64, root_weight=False, use_bn=True, dim_share=256)
    assert sorted(m.state_dict().keys()) == ref_keys
    for n, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(k["sd." + n].shape), n


def test_data_bag_and_file_roundtrip(tmp_path):
    from bridged_gnn_amd import Data, load_bridged_graph, save_bridged_graph
    d = Data(x=torch.randn(5, 3), edge_index=torch.tensor([[0, 1, 1], [1, 2, 2]]), y=torch.tensor([0, 1, -1, 2, 0]),
             train_mask=torch.tensor([1, 0, 0, 1, 0], dtype=torch.bool), central_mask=torch.tensor([1, 1, 1, 0, 0], dtype=torch.bool))
    assert d.num_nodes == 5 and d.num_features == 3 and d.num_edges == 3
    assert [k for k, _ in d("train_mask", "val_mask")] == ["train_mask"]
    p = str(tmp_path / "g_bridged_graph.dat")
    save_bridged_graph(d, p)
    e = load_bridged_graph(p)
    for key in d.keys:
        assert torch.equal(getattr(d, key), getattr(e, key))
    # the on-disk layout is the reference's (Data -> _store -> _mapping), readable by the restricted unpickler only
    import zipfile
    pk = zipfile.ZipFile(p).read([n for n in zipfile.ZipFile(p).namelist() if n.endswith("data.pkl")][0])
    assert b"torch_geometric.data.data" in pk and b"GlobalStorage" in pk and b"_mapping" in pk


def test_synthetic_generators_are_deterministic():
    from bridged_gnn_amd import synth
    a, ma = synth.bridged_graph(500, 400, 3, 5, 1000, cluster=64, seed=3)
    b, mb = synth.bridged_graph(500, 400, 3, 5, 1000, cluster=64, seed=3)
    assert np.array_equal(a, b) and np.array_equal(ma, mb)
    assert a.shape == (2, 3 * 900 + 5 * 400 + 1000)
    cross = a[:, 3 * 900: 3 * 900 + 5 * 400]
    assert (cross[0] < 500).all() and (cross[1] >= 500).all()          # bridge edges are s -> t
    x, ei, y, m = synth.sync_rd_intra(n=400, feat=8, deg=4, k_cross=3, seed=1)
    same = (y[ei[0]] == y[ei[1]])[: 400 * 4]
    assert 0.6 < same.mean() < 0.8                                      # ~70 % homophily on intra edges
