"""CPU: the N>1 path's host logic -- PartitionPlan (destination-node partition, interior/boundary
row order, halo numbering, send lists) and HaloExchange over torch.distributed -- with the gloo backend
at world_size 2 (real processes) and world_size 4 (all plans simulated in one process).  The compute
steps are done by the ORACLE (checker only): if the plan or the exchange is wrong the per-rank results
cannot reproduce the single-process oracle output."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle_c as OC
from oracle import oracle_np as O


def _problem(seed=0, n_src=700, n_tar=500, D=12):
    from bridged_gnn_amd import synth
    ei, mask = synth.bridged_graph(n_src, n_tar, k_within=3, k_cross=5, n_extra=4000, cluster=64, p_local=0.7, seed=seed)
    rng = np.random.default_rng(seed)
    n = n_src + n_tar
    h_t2s = rng.standard_normal((n, D)).astype(np.float32)
    h_s2t = rng.standard_normal((n, D)).astype(np.float32)
    a1 = rng.standard_normal(D).astype(np.float32)
    a2 = rng.standard_normal(D).astype(np.float32)
    rowptr, col, _ = O.dst_csr(ei, mask)
    ref = OC.adaptedconv_aggregate(h_t2s, h_s2t, a1, a2, rowptr, col, mask)
    return ei, mask, h_t2s, h_s2t, a1, a2, ref


def _aggregate_big(p, big, a1, a2):
    """oracle aggregation over a rank's [h_s2t local | h_t2s local | halo] allocation (col indices are relative to
    each table's base, see PartitionPlan.table_views)."""
    h_t2s, h_s2t = big[p.n_local:], big
    n = h_s2t.shape[0]
    pad = np.zeros((n - h_t2s.shape[0], big.shape[1]), np.float32)
    h_t2s = np.concatenate([h_t2s, pad])                                   # same row count for the C oracle
    rowptr = np.concatenate([p.rowptr, np.full(n - p.n_local, p.rowptr[-1], np.int32)])
    mask = np.concatenate([p.mask_local, np.zeros(n - p.n_local, bool)])
    return OC.adaptedconv_aggregate(h_t2s, h_s2t, a1, a2, rowptr, p.col, mask)[: p.n_local]


def _fill_local(p, h_t2s_full, h_s2t_full, D):
    big = np.zeros((2 * p.n_local + p.n_halo, D), np.float32)
    big[: p.n_local] = h_s2t_full[p.owned_global]
    big[p.n_local: 2 * p.n_local] = h_t2s_full[p.owned_global]
    return big


def test_plan_world4_simulated_exchange():
    from bridged_gnn_amd.dist import PartitionPlan
    ei, mask, h_t2s, h_s2t, a1, a2, ref = _problem(seed=1)
    world = 4
    plans = [PartitionPlan(ei, mask, r, world) for r in range(world)]
    assert sum(p.n_local for p in plans) == mask.shape[0]
    assert sum(p.local_num_edges for p in plans) == plans[0].global_num_edges
    D = ref.shape[1]
    bigs = [_fill_local(p, h_t2s, h_s2t, D) for p in plans]
    got = np.zeros_like(ref)
    for r, p in enumerate(plans):
        off = 2 * p.n_local
        for q, pq in enumerate(plans):                     # what q sends me, in q's send order (= all_to_all_single)
            s0 = sum(pq.send_splits[:r])
            rows = pq.send_rows[s0: s0 + pq.send_splits[r]]
            assert len(rows) == p.recv_splits[q]
            bigs[r][off: off + len(rows)] = bigs[q][rows]
            off += len(rows)
        assert off == bigs[r].shape[0]
        ri = p.rowptr[p.n_interior]
        assert (p.col[:ri] < p.n_local).all()              # interior rows never touch the halo
        # local-source / remote-source split covers every edge exactly once, in row order
        assert (p.col_L < p.n_local).all() and (p.col_R >= p.n_local).all()
        assert np.array_equal(np.diff(p.rowptr_L) + np.diff(p.rowptr_R), np.diff(p.rowptr))
        assert np.diff(p.rowptr_R)[: p.n_interior].sum() == 0 and (np.diff(p.rowptr_R)[p.n_interior:] > 0).all()
        for i in (0, p.n_interior, p.n_local - 1):
            full = p.col[p.rowptr[i]: p.rowptr[i + 1]]
            parts = np.concatenate([p.col_L[p.rowptr_L[i]: p.rowptr_L[i + 1]], p.col_R[p.rowptr_R[i]: p.rowptr_R[i + 1]]])
            assert np.array_equal(np.sort(full), np.sort(parts))
        got[p.owned_global] = _aggregate_big(p, bigs[r], a1, a2)
    assert np.array_equal(got, ref)          # same per-row arithmetic order -> bitwise equal
    # resident-input-halo layout of the first conv: two tables of n_local + n_halo rows, halo slot s at row n_local + s
    # of the table that needs it, filled from the slot's global id (what `exchange_rows` delivers)
    got2 = np.zeros_like(ref)
    for p in plans:
        ids = np.concatenate([p.owned_global, p.halo_global[p.halo_ext_perm]])     # halo regrouped: t2s-only, then s2t-only
        assert np.array_equal(p.halo_mask, mask[p.halo_global])
        t2s_e, s2t_e = h_t2s[ids].copy(), h_s2t[ids].copy()
        n0, n1 = p.n_halo_by_table
        s2t_e[p.n_local: p.n_local + n0] = np.nan                                # tables a halo group does NOT need may
        t2s_e[p.n_local + n0:] = np.nan                                          # stay unwritten: poison them
        n = ids.shape[0]
        rowptr = np.concatenate([p.rowptr, np.full(n - p.n_local, p.rowptr[-1], np.int32)])
        m_ext = np.concatenate([p.mask_local, p.halo_mask[p.halo_ext_perm]])
        got2[p.owned_global] = OC.adaptedconv_aggregate(t2s_e, s2t_e, a1, a2, rowptr, p.col_ext, m_ext)[: p.n_local]
        # what the owners would send for the input features: plain local row numbers of the same send list
        for q, pq in enumerate(plans):
            assert np.array_equal(pq.owned_global[pq.send_rows_local], pq.owned_global[pq.send_rows % pq.n_local])
    assert np.array_equal(got2, ref)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bridged_gnn_amd.dist import HaloExchange, PartitionPlan
        ei, mask, h_t2s, h_s2t, a1, a2, ref = _problem(seed=2)
        p = PartitionPlan(ei, mask, rank, world)
        D = ref.shape[1]
        big = torch.from_numpy(_fill_local(p, h_t2s, h_s2t, D))
        hx = HaloExchange(p, "cpu")
        hx.start(big)
        hx.wait()
        out = _aggregate_big(p, big.numpy(), a1, a2)
        ok = bool(np.array_equal(out, ref[p.owned_global]))
        # SURVEY 8(e)'s fallback: the same halo out of ONE all_gather of every rank's [2 n_local, D] block
        big2 = torch.from_numpy(_fill_local(p, h_t2s, h_s2t, D))
        hg = HaloExchange(p, "cpu", mode="allgather")
        hg.start(big2)
        hg.wait()
        ok = ok and bool(np.array_equal(big2.numpy(), big.numpy())) and hg.mode == "allgather"
        ok = ok and HaloExchange(p, "cpu").mode == ("allgather" if p.halo_fraction >= HaloExchange.ALLGATHER_FROM else "a2a")
        # input-feature halo (fetched once per version of x): owners send plain local rows, slots arrive in halo order
        feat = np.random.default_rng(9).standard_normal((mask.shape[0], 8)).astype(np.float32)
        mine = torch.from_numpy(feat[p.owned_global])
        halo = hx.exchange_rows(mine.index_select(0, torch.from_numpy(p.send_rows_local)))
        ok = ok and bool(np.array_equal(halo.numpy(), feat[p.halo_global]))
        # (1) of the per-conv protocol: per-domain sums are all-reducible
        x = np.random.default_rng(5).standard_normal((mask.shape[0], 6))
        loc = x[p.owned_global]
        sums = torch.tensor(np.concatenate([loc[p.mask_local].sum(0), loc[~p.mask_local].sum(0),
                                            [p.mask_local.sum(), (~p.mask_local).sum()]]))
        dist.all_reduce(sums)
        glob = np.concatenate([x[mask].sum(0), x[~mask].sum(0), [mask.sum(), (~mask).sum()]])
        ok = ok and bool(np.allclose(sums.numpy(), glob, rtol=1e-12))
        q.put((rank, ok, p.summary()))
    finally:
        dist.destroy_process_group()


def test_halo_exchange_gloo_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(s["n_halo"] > 0 and min(s["n_halo_by_table"]) > 0 for _, _, s in res)      # the exchange was exercised


def test_partition_schemes():
    from bridged_gnn_amd.dist import partition_nodes
    m = np.array([True] * 10 + [False] * 6)
    o = partition_nodes(m, 4)
    assert np.bincount(o[:10], minlength=4).tolist() == [3, 2, 3, 2] and np.bincount(o[10:], minlength=4).tolist() == [2, 1, 2, 1]
    assert (np.diff(o[:10]) >= 0).all() and (np.diff(o[10:]) >= 0).all()
    o2 = partition_nodes(m, 3, "contiguous")
    assert (np.diff(o2) >= 0).all() and set(o2) == {0, 1, 2}


# ------------------------------------------------------------------------------------------------ kNN bridge, sharded
class _OracleKnnBackend:
    """compute stand-in for the HIP ops in the CPU test of the sharded kNN HOST logic (sharding, the all_gather of the
    candidates, global ids, edge assembly): the oracle's canonical routines on CPU tensors"""

    @staticmethod
    def l2_normalize_rows(q):
        return torch.from_numpy(OC.l2_normalize_rows(q.numpy()))

    @staticmethod
    def cosine_topk(qn, cn, k, apply_sigmoid=True):
        val, idx = OC.cosine_topk(qn.numpy(), cn.numpy(), k)
        v = O.sigmoid_f32(val) if apply_sigmoid else val.astype(np.float32)
        return torch.from_numpy(idx), torch.from_numpy(v), torch.zeros(2, dtype=torch.int32)

    @staticmethod
    def topk_edges(idx, cand_base=0, query_base=0):
        nq, k = idx.shape
        to = torch.arange(nq, dtype=torch.int64).repeat_interleave(k) + query_base
        return torch.stack([idx.reshape(-1) + cand_base, to])

    @staticmethod
    def coalesce(ei):
        return torch.from_numpy(O.coalesce(ei.numpy()))


def _knn_problem():
    from bridged_gnn_amd import synth
    q = synth.gaussian_embeddings(301, 32, seed=3)          # 301 / 1000: uneven shards at world 2 and 3
    c = synth.gaussian_embeddings(1000, 32, seed=4)
    c[500:520] = c[100:120]                                  # exact duplicates: ties are broken by the lower index
    return q, c


def _knn_worker(rank, world, port, q_out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bridged_gnn_amd import bridge
        from bridged_gnn_amd.dist import shard_range
        q, c = _knn_problem()
        qlo, qhi = shard_range(q.shape[0], rank, world)
        clo, chi = shard_range(c.shape[0], rank, world)
        ei, idx, val, _ = bridge.sharded_cosine_topk_edges(torch.from_numpy(q[qlo:qhi]), torch.from_numpy(c[clo:chi]), 7,
                                                           rank=rank, world=world, query_base=qlo, backend=_OracleKnnBackend)
        full = bridge.gather_edges(ei, world=world, backend=_OracleKnnBackend)
        q_out.put((rank, ei.numpy(), idx.numpy(), val.numpy(), full.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_knn_union_equals_single_rank(world):
    """row N2 / SURVEY 8(e): query-row shards + ONE all_gather of the candidates; the union of the per-rank edge lists
    (and the gathered list every rank ends up with) is bit-identical to the single-rank coalesced edge list."""
    from bridged_gnn_amd import bridge
    q, c = _knn_problem()
    one_ei, one_idx, one_val, _ = bridge.sharded_cosine_topk_edges(torch.from_numpy(q), torch.from_numpy(c), 7,
                                                                   backend=_OracleKnnBackend)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    qq = ctx.Queue()
    procs = [ctx.Process(target=_knn_worker, args=(r, world, port, qq)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([qq.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(np.concatenate([r[2] for r in res]), one_idx.numpy())          # [Nq/P, k] slices in rank order
    assert np.array_equal(np.concatenate([r[3] for r in res]), one_val.numpy())
    union = O.coalesce(np.concatenate([r[1] for r in res], axis=1))
    assert np.array_equal(union, one_ei.numpy())
    assert sum(r[1].shape[1] for r in res) == one_ei.shape[1]                            # no edge on two ranks
    for r in res:
        assert np.array_equal(r[4], one_ei.numpy())                                      # gather_edges: whole on every rank
