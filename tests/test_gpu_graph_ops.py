"""GPU: integer edge-list plumbing through the C ABI, bit-exact vs the oracle / golden vectors."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_csr_office_bit_exact(golden):
    from bridged_gnn_amd import ops
    g, p = golden("office_a2d_graph.npz"), golden("partition_office.npz")
    und = p["ei_undirected"].astype(np.int64)
    csr = ops.build_dst_csr(_t(und), g["x"].shape[0], rewrite_self_loops=True, want_eperm=True)
    rowptr, col, eperm = O.dst_csr(und, g["central_mask"])
    assert csr.num_edges == 37522
    assert np.array_equal(csr.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(csr.col.cpu().numpy(), col)
    assert np.array_equal(csr.eperm.cpu().numpy(), eperm)


@pytest.mark.parametrize("n,e,seed", [(200, 1500, 1), (1000, 0, 2), (5000, 60000, 3), (1, 3, 4)])
def test_csr_random_multigraph(n, e, seed):
    from bridged_gnn_amd import ops, synth
    ei, mask = synth.random_multigraph(max(n, 2), e, n_isolated=min(7, n // 4), seed=seed) if e else \
        (np.zeros((2, 0), np.int64), np.arange(n) % 2 == 0)
    n = mask.shape[0]
    csr = ops.build_dst_csr(_t(ei), n, rewrite_self_loops=True, want_eperm=True)
    rowptr, col, eperm = O.dst_csr(ei, mask)
    assert np.array_equal(csr.rowptr.cpu().numpy(), rowptr)
    assert np.array_equal(csr.col.cpu().numpy(), col)
    assert np.array_equal(csr.eperm.cpu().numpy(), eperm)
    # rewrite off: self loops kept as ordinary edges, no loops appended
    csr2 = ops.build_dst_csr(_t(ei), n, rewrite_self_loops=False)
    assert csr2.num_edges == ei.shape[1]
    if ei.shape[1]:
        order = np.argsort(ei[1], kind="stable")
        assert np.array_equal(csr2.col.cpu().numpy(), ei[0][order].astype(np.int32))


def test_coalesce_and_undirected(golden):
    from bridged_gnn_amd import utils
    g, p = golden("office_a2d_graph.npz"), golden("partition_office.npz")
    ei = g["edge_index"].astype(np.int64)
    und = utils.to_undirected(_t(ei), g["x"].shape[0]).cpu().numpy()
    assert np.array_equal(und, p["ei_undirected"])
    rng = np.random.default_rng(0)
    for n, e in ((50, 400), (100000, 300000), (7, 1)):
        r = rng.integers(0, n, size=(2, e))
        got = utils.coalesce(_t(r), num_nodes=n).cpu().numpy()
        assert np.array_equal(got, O.coalesce(r, n))
        got2 = utils.coalesce(_t(r)).cpu().numpy()                 # num_nodes inferred = max+1
        assert np.array_equal(got2, O.coalesce(r))
    assert utils.coalesce(torch.zeros(2, 0, dtype=torch.int64, device=DEV)).shape == (2, 0)


def test_graph_partition_api(golden):
    from bridged_gnn_amd.ktgnn import KTGNN_no_complement
    g, p = golden("office_a2d_graph.npz"), golden("partition_office.npz")
    m = KTGNN_no_complement(256, 31, 2, 64, dim_share=256)
    e1, e2, cat = m.graph_partition(_t(p["ei_undirected"].astype(np.int64)), _t(g["central_mask"]))
    assert np.array_equal(e1.cpu().numpy(), p["e1"]) and np.array_equal(e2.cpu().numpy(), p["e2"])
    assert cat.shape[1] == 37522


def test_topk_edges_and_merge():
    from bridged_gnn_amd import ops
    from bridged_gnn_amd.bridge import merge_graphs
    from bridged_gnn_amd.data import Data
    idx = torch.tensor([[2, 0], [1, 1], [0, 2]], dtype=torch.int64, device=DEV)
    e = ops.topk_edges(idx).cpu().numpy()
    assert e.tolist() == [[2, 0, 1, 1, 0, 2], [0, 0, 1, 1, 2, 2]]
    assert np.array_equal(ops.coalesce(_t(e)).cpu().numpy(), O.topk_edges(idx.cpu().numpy()))
    ds = Data(x=torch.zeros(3, 2, device=DEV), edge_index=_t(np.array([[0, 1], [1, 2]])), y=torch.tensor([0, -1, 1], device=DEV))
    dt = Data(x=torch.ones(2, 2, device=DEV), edge_index=_t(np.array([[0], [1]])), y=torch.tensor([1, 0], device=DEV),
              train_mask=torch.tensor([True, False], device=DEV), val_mask=torch.tensor([False, True], device=DEV),
              test_mask=torch.tensor([False, False], device=DEV))
    cross = _t(np.array([[2, 2], [0, 0]]))
    m = merge_graphs(ds, dt, cross)
    assert m.edge_index.cpu().tolist() == O.merge_graphs(3, 2, [[0, 1], [1, 2]], [[0], [1]], [[2, 2], [0, 0]]).tolist()
    assert cross.cpu().tolist() == [[2, 2], [0, 0]]                      # argument untouched
    assert m.central_mask.cpu().tolist() == [True, True, True, False, False]
    assert m.train_mask.cpu().tolist() == [True, False, True, True, False]
    assert m.val_mask.cpu().tolist() == [False, False, False, False, True]


def test_cpu_tensors_are_refused():
    from bridged_gnn_amd import ops, utils
    with pytest.raises(RuntimeError):
        ops.build_dst_csr(torch.zeros(2, 3, dtype=torch.int64), 4)
    with pytest.raises(RuntimeError):
        utils.coalesce(torch.zeros(2, 3, dtype=torch.int64))


def test_training_entry_points_reject_shapes_outside_their_envelope():
    """argument validation of the dense-side entry points (no kernel runs): the callers rely on these codes to fall
    back to a library GEMM outside the kernels' envelopes."""
    from bridged_gnn_amd import _lib
    L = _lib.lib()
    x = torch.zeros(64, 128, device=DEV)
    w = torch.zeros(96, 128, device=DEV)
    b = torch.zeros(96, device=DEV)
    o = torch.zeros(64, 96, device=DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    # Dout % 64 != 0
    assert L.bgnn_linear_f32(p(x), 64, 128, 128, p(w), p(b), 96, 0, None, None, p(o), 96, None) == -2
    # Din > 128
    assert L.bgnn_linear_f32(p(x), 64, 256, 256, p(w), p(b), 64, 0, None, None, p(o), 64, None) == -2
    # column sums without a mask
    s = torch.zeros(130, dtype=torch.float64, device=DEV)
    assert L.bgnn_linear_f32(p(x), 64, 128, 128, p(w), p(b), 64, 0, None, p(s), p(o), 64, None) == -1
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device=DEV)
    # gram: p > 288, q > 128, q % 4 != 0, workspace too small
    assert L.bgnn_gram_f32(p(x), 128, 292, p(x), 128, 128, 64, p(o), p(ws), ws.numel(), None) == -2
    assert L.bgnn_gram_f32(p(x), 128, 128, p(x), 128, 132, 64, p(o), p(ws), ws.numel(), None) == -2
    assert L.bgnn_gram_f32(p(x), 128, 128, p(x), 128, 6, 64, p(o), p(ws), ws.numel(), None) == -2
    assert L.bgnn_gram_f32(p(x), 128, 128, p(x), 128, 128, 64, p(o), p(ws), 16, None) == -3
    # rowdot: more than four vectors, d > 256
    assert L.bgnn_rowdot_f32(p(x), 128, 64, 128, p(w), 128, 5, p(o), None) == -2
    assert L.bgnn_rowdot_f32(p(x), 128, 64, 260, p(w), 260, 1, p(o), None) == -2
    # pull backward: D > 128 has no pull form (the record holds 4 x 32 sign bits per edge)
    i32 = torch.zeros(8, dtype=torch.int32, device=DEV)
    m = torch.zeros(8, dtype=torch.uint8, device=DEV)
    rc = L.bgnn_adaptedconv_aggregate_bwd_pull_f32(p(x), p(x), 132, p(b), p(b), p(i32), p(i32), p(m), p(i32), p(i32), p(i32), 4, 0, 132, 0.1,
                                                   p(x), 132, p(b), p(x), 132, p(o), p(o), p(b), p(b), p(ws), ws.numel(), None)
    assert rc == -2



@pytest.mark.parametrize("w,ld", [(12, 12), (128, 128), (4, 8), (36, 40)])
def test_gather_rows_matches_index_select(w, ld):
    from bridged_gnn_amd import ops
    g = torch.Generator(device=DEV).manual_seed(w)
    table = torch.randn(5000, ld, device=DEV, generator=g)[:, :w]          # row-strided view when ld > w
    idx = torch.randint(0, 5000, (12345,), device=DEV, generator=g)
    assert torch.equal(ops.gather_rows(table, idx), table.index_select(0, idx))
    assert ops.gather_rows(table, idx[:0]).shape == (0, w)
    with pytest.raises(RuntimeError):
        ops.gather_rows(table.t().contiguous().t(), idx)                   # column-major view: no unit column stride


@pytest.mark.parametrize("nq,nc,k,seed", [(1000, 700, 20, 0), (257, 5000, 3, 1), (4, 4, 4, 2), (100000, 100000, 20, 3), (1, 9, 1, 4)])
def test_topk_edges_coalesced_equals_coalesce_of_topk_edges(nq, nc, k, seed):
    """bgnn_topk_edges_coalesced_i64 (one stable pair sort) == bgnn_coalesce_i64(bgnn_topk_edges_i64(...)) bit for bit, with
    bases (main_bridged_graph.py:61-68,:75)."""
    from bridged_gnn_amd import ops
    g = torch.Generator(device=DEV).manual_seed(seed)
    # k distinct candidates per query, in arbitrary (top-k) order
    idx = torch.rand(nq, nc, device=DEV, generator=g).topk(k, dim=1).indices.contiguous() if nq * nc <= 50_000_000 else \
        (torch.randint(0, nc - k, (nq, 1), device=DEV, generator=g) + torch.randperm(k, device=DEV, generator=g)[None, :]).contiguous()
    for cb, qb in ((0, 0), (7, 123)):
        want = ops.coalesce(ops.topk_edges(idx, cand_base=cb, query_base=qb), num_nodes=max(nq + qb, nc + cb))
        got = ops.topk_edges_coalesced(idx, nc, cand_base=cb, query_base=qb)
        assert got.shape == want.shape and torch.equal(got, want)
