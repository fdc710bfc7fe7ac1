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


def _local_aggregate(plan, tabs, a1, a2):
    return OC.adaptedconv_aggregate(tabs[0], tabs[1], a1, a2, plan.rowptr, plan.col,
                                    np.concatenate([plan.mask_local, np.zeros(max(tabs[0].shape[0], tabs[1].shape[0]) - plan.n_local, bool)]))[: plan.n_local]


def test_plan_world4_simulated_exchange():
    from bridged_gnn_amd.dist import PartitionPlan
    ei, mask, h_t2s, h_s2t, a1, a2, ref = _problem(seed=1)
    world = 4
    plans = [PartitionPlan(ei, mask, r, world) for r in range(world)]
    assert sum(p.n_local for p in plans) == mask.shape[0]
    assert sum(p.local_num_edges for p in plans) == plans[0].global_num_edges
    full = (h_t2s, h_s2t)
    got = np.zeros_like(ref)
    for r, p in enumerate(plans):
        tabs = []
        for t in (0, 1):
            tab = np.zeros((p.n_local + p.n_halo[t], ref.shape[1]), np.float32)
            tab[: p.n_local] = full[t][p.owned_global]
            off = p.n_local
            for q, pq in enumerate(plans):                     # what q sends me, in q's send order
                s0 = sum(pq.send_splits[t][:r])
                rows = pq.send_rows[t][s0: s0 + pq.send_splits[t][r]]
                assert len(rows) == p.recv_splits[t][q]
                tab[off: off + len(rows)] = full[t][pq.owned_global[rows]]
                off += len(rows)
            assert off == tab.shape[0]
            tabs.append(tab)
        # interior rows must not reference halo rows
        ri = p.rowptr[p.n_interior]
        assert (p.col[:ri] < p.n_local).all()
        m2 = p.mask_local
        out = OC.adaptedconv_aggregate(_pad_rows(tabs[0], tabs[1])[0], _pad_rows(tabs[0], tabs[1])[1], a1, a2,
                                       np.concatenate([p.rowptr, np.full(_extra(tabs, p), p.rowptr[-1], np.int32)]),
                                       p.col, np.concatenate([m2, np.zeros(_extra(tabs, p), bool)]))[: p.n_local]
        got[p.owned_global] = out
    assert np.array_equal(got, ref)          # same per-row arithmetic order -> bitwise equal


def _extra(tabs, p):
    return max(tabs[0].shape[0], tabs[1].shape[0]) - p.n_local


def _pad_rows(a, b):
    n = max(a.shape[0], b.shape[0])
    pa = np.zeros((n, a.shape[1]), np.float32); pa[: a.shape[0]] = a
    pb = np.zeros((n, b.shape[1]), np.float32); pb[: b.shape[0]] = b
    return pa, pb


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bridged_gnn_amd.dist import HaloExchange, PartitionPlan
        ei, mask, h_t2s, h_s2t, a1, a2, ref = _problem(seed=2)
        p = PartitionPlan(ei, mask, rank, world)
        D = ref.shape[1]
        tabs = []
        for t, full in enumerate((h_t2s, h_s2t)):
            tab = torch.zeros(p.n_local + p.n_halo[t], D)
            tab[: p.n_local] = torch.from_numpy(full[p.owned_global])
            tabs.append(tab)
        hx = HaloExchange(p, "cpu")
        hx.start(tabs)
        hx.wait()
        pa, pb = _pad_rows(tabs[0].numpy(), tabs[1].numpy())
        ex = _extra([pa, pb], p) if False else pa.shape[0] - p.n_local
        out = OC.adaptedconv_aggregate(pa, pb, a1, a2, np.concatenate([p.rowptr, np.full(ex, p.rowptr[-1], np.int32)]),
                                       p.col, np.concatenate([p.mask_local, np.zeros(ex, bool)]))[: p.n_local]
        ok = bool(np.array_equal(out, ref[p.owned_global]))
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
    assert all(s["n_halo"][0] + s["n_halo"][1] > 0 for _, _, s in res)      # the exchange was exercised


def test_partition_schemes():
    from bridged_gnn_amd.dist import partition_nodes
    m = np.array([True] * 10 + [False] * 6)
    o = partition_nodes(m, 4)
    assert np.bincount(o[:10], minlength=4).tolist() == [3, 2, 3, 2] and np.bincount(o[10:], minlength=4).tolist() == [2, 1, 2, 1]
    assert (np.diff(o[:10]) >= 0).all() and (np.diff(o[10:]) >= 0).all()
    o2 = partition_nodes(m, 3, "contiguous")
    assert (np.diff(o2) >= 0).all() and set(o2) == {0, 1, 2}
