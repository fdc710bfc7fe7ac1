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


def _csr_cpu(ei, n_dst):
    """by-destination CSR of an edge list on CPU tensors (host logic only: what bgnn_build_dst_csr returns, without self-loop rewriting)"""
    import numpy as np
    import torch
    from bridged_gnn_amd.ops import DstCSR
    order = np.argsort(ei[1], kind="stable")
    col = ei[0][order].astype(np.int32)
    rowptr = np.zeros(n_dst + 1, dtype=np.int32)
    np.add.at(rowptr, ei[1] + 1, 1)
    rowptr = np.cumsum(rowptr).astype(np.int32)
    return DstCSR(torch.from_numpy(rowptr), torch.from_numpy(col), None, col.shape[0], n_dst)


def test_tile_need_over_own_and_halo_rows_matches_brute_force():
    """`DstCSR.tile_need(mask, table_mask_u8=...)` (a rank's graph: destinations = own rows, tables = own rows followed by halo rows): bit 0 / 1 of
    a 32-row tile = some destination of the target / source domain reads one of its rows from h_s2t / h_t2s, an own row's own table included."""
    import numpy as np
    import torch
    rng = np.random.default_rng(0)
    n_dst, n_halo, E = 333, 150, 2500
    R = n_dst + n_halo
    ei = np.stack([rng.integers(0, R, E), rng.integers(0, n_dst, E)])
    ei = ei[:, ~((ei[0] >= n_dst + 100))]                       # the last 50 halo rows are referenced by nobody
    mask = rng.random(n_dst) < 0.4
    tmask = np.concatenate([mask, rng.random(n_halo) < 0.5])
    csr = _csr_cpu(ei, n_dst)
    need = csr.tile_need(torch.from_numpy(mask).to(torch.uint8), table_mask_u8=torch.from_numpy(tmask).to(torch.uint8))
    assert need is not None and need.shape[0] == (R + 31) // 32
    want = np.zeros((R + 31) // 32, dtype=np.int32)
    for i in range(n_dst):
        want[i // 32] |= 2 if mask[i] else 1                    # the row's own table (the logit reads h_i)
    for j, i in ei.T:
        want[j // 32] |= 2 if mask[i] else 1                    # source-domain destinations gather from h_t2s (bit 1), targets from h_s2t (bit 0)
    assert np.array_equal(need.numpy(), want)
    assert (need.numpy()[-1] == 0)                              # nobody reads the unreferenced halo tail
    # square graph, every row read from both tables -> None
    full = np.stack([np.repeat(np.arange(64), 2), np.tile(np.array([0, 40]), 64)])
    csr2 = _csr_cpu(full, 64)
    m2 = np.zeros(64, dtype=bool); m2[:32] = True
    assert csr2.tile_need(torch.from_numpy(m2).to(torch.uint8)) is None


def test_gather_hint_separates_clustered_from_scattered_graphs():
    """`DstCSR.gather_hint()`: 1 when neighbouring destination rows share neighbours (bridged / kNN graphs), 2 when they do not."""
    import numpy as np
    rng = np.random.default_rng(1)
    n, deg = 200_000, 8
    dst = np.repeat(np.arange(n), deg)
    clustered = np.stack([(dst // 512) * 512 + rng.integers(0, 512, n * deg), dst])          # neighbours from the row's own block of 512
    scattered = np.stack([rng.integers(0, n, n * deg), dst])
    assert _csr_cpu(clustered, n).gather_hint() == 1
    assert _csr_cpu(scattered, n).gather_hint() == 2
    assert _csr_cpu(np.stack([rng.integers(0, 100, 800), np.repeat(np.arange(100), 8)]), 100).gather_hint() == 1      # too small to matter: default
