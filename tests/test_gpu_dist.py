"""GPU: the partitioned driver's device pieces (row-range aggregation over local+halo tables, transform
into oversized tables, world=1 PartitionedKTGNN).  RCCL itself needs >1 GPU (driver-side scaling run);
here the exchange is simulated in-process by copying rows exactly as all_to_all_single would deliver them."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _model(feat, hidden, classes):
    from bridged_gnn_amd.ktgnn import KTGNN_no_complement
    torch.manual_seed(0)
    m = KTGNN_no_complement(feat, classes, 2, hidden, use_bn=True, dim_share=feat)
    g = torch.Generator().manual_seed(3)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
    return m.to(DEV).eval()


def test_partitioned_world1_equals_plain_forward():
    from bridged_gnn_amd import synth
    from bridged_gnn_amd.data import Data
    from bridged_gnn_amd.dist import PartitionedKTGNN
    ei, mask = synth.bridged_graph(3000, 2000, 4, 8, 6000, cluster=128, seed=4)
    x = torch.randn(5000, 64, device=DEV)
    m = _model(64, 64, 3)
    with torch.no_grad():
        ref = m(Data(x=x, edge_index=_t(ei), central_mask=_t(mask)))[:3]
    pk = PartitionedKTGNN(m, ei, mask, 0, 1, DEV)
    out = pk.forward(x[pk.owned_global])
    inv = torch.empty_like(pk.owned_global)
    inv[pk.owned_global] = torch.arange(5000, device=DEV)
    for a, b in zip(out, ref):
        assert torch.allclose(a[inv], b, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_simulated_multi_rank_conv_matches_single_gpu(world):
    """every rank's (transform -> exchanged halo -> interior/boundary aggregation) reproduces the
    single-GPU AdaptedConv output row for row."""
    from bridged_gnn_amd import ops, synth
    from bridged_gnn_amd.dist import PartitionPlan
    from bridged_gnn_amd.ktgnn import AdaptedConv, _pad_cols4
    n_src, n_tar, din, D = 4000, 3000, 32, 64
    ei, mask = synth.bridged_graph(n_src, n_tar, 4, 8, 9000, cluster=128, p_local=0.8, seed=world)
    n = n_src + n_tar
    x = torch.randn(n, din, device=DEV)
    torch.manual_seed(1)
    conv = AdaptedConv(din, D, root_weight=False).to(DEV).eval()
    with torch.no_grad():
        csr = ops.build_dst_csr(_t(ei), n)
        ref = conv(x, None, central_mask=_t(mask), csr=csr)
    plans = [PartitionPlan(ei, mask, r, world) for r in range(world)]
    # (1) sums: sum over ranks of local sums == global sums (the all-reduce)
    locs = []
    for p in plans:
        og = _t(p.owned_global)
        locs.append(ops.domain_sums(_pad_cols4(x[og]), _t(p.mask_local).to(torch.uint8)))
    sums = torch.stack(locs).sum(0)
    delta = ops.domain_delta(sums, din)
    # (2) per-rank transform straight into the per-conv allocation [h_s2t local | h_t2s local | halo]
    bigs = []
    for p in plans:
        og = _t(p.owned_global)
        big = torch.zeros(2 * p.n_local + p.n_halo, ops.pad4(D), device=DEV)
        with torch.no_grad():
            conv.transform(x[og].contiguous(), _t(p.mask_local).to(torch.uint8), delta=delta, out=p.table_views(big))
        bigs.append(big)
    # (3) simulated all_to_all_single: receiver r gets, peer by peer, what q's send list holds for r
    for r, p in enumerate(plans):
        off = 2 * p.n_local
        for q, pq in enumerate(plans):
            s0 = sum(pq.send_splits[:r])
            rows = pq.send_rows[s0: s0 + pq.send_splits[r]]
            assert len(rows) == p.recv_splits[q]
            if len(rows):
                bigs[r][off: off + len(rows)] = bigs[q][_t(rows)]
            off += len(rows)
        assert off == bigs[r].shape[0]
    # (4) interior then boundary aggregation per rank
    got = torch.zeros_like(ref)
    a_t2s = conv.a_f_t2s.weight.detach().reshape(-1).contiguous()
    a_s2t = conv.a_f_s2t.weight.detach().reshape(-1).contiguous()
    for r, p in enumerate(plans):
        lcsr = ops.DstCSR(_t(p.rowptr), _t(p.col), None, p.local_num_edges, p.n_local)
        m8 = _t(p.mask_local).to(torch.uint8)
        h_t2s, h_s2t = p.table_views(bigs[r])
        out = torch.full((p.n_local, ops.pad4(D)), float("nan"), device=DEV)
        ops.adaptedconv_aggregate(h_t2s, h_s2t, a_t2s, a_s2t, lcsr, m8, D, n_dst=p.n_local, out=out,
                                  row_begin=0, row_end=p.n_interior)
        assert torch.isnan(out[p.n_interior:]).all() or p.n_interior == p.n_local     # only the range was written
        ops.adaptedconv_aggregate(h_t2s, h_s2t, a_t2s, a_s2t, lcsr, m8, D, n_dst=p.n_local, out=out,
                                  row_begin=p.n_interior, row_end=p.n_local)
        got[_t(p.owned_global)] = out[:, :D]
    assert torch.allclose(got, ref, rtol=1e-6, atol=1e-6)
    assert sum(p.n_halo for p in plans) > 0


def _gloo_gpu_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bridged_gnn_amd import synth
        from bridged_gnn_amd.data import Data
        from bridged_gnn_amd.dist import PartitionedKTGNN
        ei, mask = synth.bridged_graph(3000, 2000, 4, 8, 6000, cluster=128, p_local=0.8, seed=4)
        m = _model(64, 64, 3)
        g = torch.Generator(device=DEV).manual_seed(1)
        x = torch.randn(5000, 64, device=DEV, generator=g)
        with torch.no_grad():
            ref = m(Data(x=x, edge_index=_t(ei), central_mask=_t(mask)))[:3]
        ok = True
        # resident input halo for the first conv / exchange for every conv / the all-gather fallback of SURVEY 8(e)
        for cache, mode in ((True, "auto"), (False, "auto"), (False, "allgather")):
            pk = PartitionedKTGNN(m, ei, mask, rank, world, DEV, cache_input_halo=cache, halo_mode=mode)
            xl = x[pk.owned_global].contiguous()
            out = pk.forward(xl)
            ok = ok and all(torch.allclose(a, b[pk.owned_global], rtol=1e-5, atol=1e-6) for a, b in zip(out, ref))
            if cache:                            # an in-place update of the features must re-fetch their halo
                xl.mul_(0.5)
                with torch.no_grad():
                    ref2 = m(Data(x=x * 0.5, edge_index=_t(ei), central_mask=_t(mask)))[:3]
                out2 = pk.forward(xl)
                ok = ok and all(torch.allclose(a, b[pk.owned_global], rtol=1e-5, atol=1e-6) for a, b in zip(out2, ref2))
        q.put((rank, bool(ok), pk.plan.summary()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_partitioned_forward_real_processes_sharing_the_gpu(world):
    """The whole partitioned driver (plan, transform into the per-conv allocation, exchange, interior/boundary launches)
    with REAL ranks.  RCCL refuses two ranks on one device, so the process group is gloo and the payload is staged through
    the host (HaloExchange.host_staging); kernels, layouts and launch order are the production ones."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(s["n_halo"] > 0 for _, _, s in res)


# ------------------------------------------------------------------------------------------------ kNN bridge, sharded
def _knn_gpu_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bridged_gnn_amd import bridge, synth
        from bridged_gnn_amd.dist import shard_range
        qe, ce = synth.gaussian_embeddings(3001, 128, seed=3), synth.gaussian_embeddings(9000, 128, seed=4)
        ce[4000:4040] = ce[100:140]                              # exact duplicates
        qlo, qhi = shard_range(3001, rank, world)
        clo, chi = shard_range(9000, rank, world)
        ei, idx, val, nfb = bridge.sharded_cosine_topk_edges(_t(qe[qlo:qhi]), _t(ce[clo:chi]), 20, rank=rank, world=world,
                                                             query_base=qlo)
        full = bridge.gather_edges(ei, world=world)
        q.put((rank, ei.cpu().numpy(), idx.cpu().numpy(), full.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_knn_real_processes_sharing_the_gpu(world):
    """row N2: the production kernels behind `bridge.sharded_cosine_topk_edges` in REAL ranks (gloo group, candidates
    all-gathered through the host because RCCL refuses two ranks per device): the union of the per-rank edge lists and the
    gathered list are bit-identical to the single-rank result and to the oracle."""
    import socket
    import torch.multiprocessing as mp
    from bridged_gnn_amd import bridge, synth
    from oracle import oracle_c as OC
    from oracle import oracle_np as O
    qe, ce = synth.gaussian_embeddings(3001, 128, seed=3), synth.gaussian_embeddings(9000, 128, seed=4)
    ce[4000:4040] = ce[100:140]
    one_ei, one_idx, _, _ = bridge.sharded_cosine_topk_edges(_t(qe), _t(ce), 20)
    _, ref_idx = OC.cosine_topk(OC.l2_normalize_rows(qe), OC.l2_normalize_rows(ce), 20)
    assert np.array_equal(one_idx.cpu().numpy(), ref_idx)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    qq = ctx.Queue()
    procs = [ctx.Process(target=_knn_gpu_worker, args=(r, world, port, qq)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([qq.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(np.concatenate([r[2] for r in res]), ref_idx)
    assert np.array_equal(O.coalesce(np.concatenate([r[1] for r in res], axis=1)), one_ei.cpu().numpy())
    for r in res:
        assert np.array_equal(r[3], one_ei.cpu().numpy())


def test_sharded_scorer_topk_slices_equal_the_whole(golden):
    """`BridgeScorer.topk(rank, world)` (mlp scorer, office ckpt): the rank slices of the [Nq, k] tables, computed one
    after the other in this process without a process group (world_size 1 collectives are identities), tile the
    single-rank tables when the candidate terms are taken whole."""
    from bridged_gnn_amd.bridge import BridgeScorer
    from bridged_gnn_amd.dist import shard_range
    from conftest import sub
    kf = golden("knn_office_a2d.npz")
    sd = {"source_learner.sim_net." + k: torch.from_numpy(np.asarray(v)) for k, v in sub(kf, "sim.").items()}
    sc = BridgeScorer(sd, DEV)
    zs, zt = _t(kf["z_src"]), _t(kf["z_tar"])
    idx, probs, _ = sc.topk(zs, zt, 20)
    for world in (2, 3):
        parts = []
        for r in range(world):
            lo, hi = shard_range(zt.shape[0], r, world)
            i, p, _ = sc.topk(zs, zt[lo:hi].contiguous(), 20)          # a rank's query slice against ALL candidates
            parts.append((i, p))
        assert torch.equal(torch.cat([i for i, _ in parts]), idx)
        # the per-node GEMM of the query terms runs on another batch shape (another library kernel): last-ulp differences
        assert torch.allclose(torch.cat([p for _, p in parts]), probs, rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------------ RCCL at world size 1
def _nccl_world1_worker(port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        from bridged_gnn_amd import synth
        from bridged_gnn_amd.data import Data
        from bridged_gnn_amd.dist import PartitionedKTGNN, all_gather_rows
        ei, mask = synth.bridged_graph(3000, 2000, 4, 8, 6000, cluster=128, p_local=0.8, seed=4)
        m = _model(64, 64, 3)
        g = torch.Generator(device=DEV).manual_seed(1)
        x = torch.randn(5000, 64, device=DEV, generator=g)
        with torch.no_grad():
            ref = m(Data(x=x, edge_index=_t(ei), central_mask=_t(mask)))[:3]
        res = {}
        for cache in (True, False):
            pk = PartitionedKTGNN(m, ei, mask, 0, 1, DEV, always_communicate=True, cache_input_halo=cache)
            xl = x[pk.owned_global].contiguous()
            with torch.no_grad():
                out = pk.forward(xl)
                res[f"eager_cache{int(cache)}"] = all(torch.allclose(a, b[pk.owned_global], rtol=1e-5, atol=1e-6) for a, b in zip(out, ref))
                # the bench's second phase: the forward INCLUDING its RCCL calls captured into a HIP graph and replayed
                for _ in range(2):
                    pk.forward(xl)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                    got = pk.forward(xl)
                for _ in range(3):
                    gr.replay()
                torch.cuda.synchronize()
                res[f"replay_cache{int(cache)}"] = all(torch.allclose(a, b[pk.owned_global], rtol=1e-5, atol=1e-6) for a, b in zip(got, ref))
        # the kNN bridge's collective on device buffers
        t = torch.randn(37, 128, device=DEV)
        res["all_gather_rows"] = bool(torch.equal(all_gather_rows(t, always=True), t))
        q.put(res)
    finally:
        dist.destroy_process_group()


def test_rccl_calls_at_world_size_1_eager_and_captured():
    """The RCCL entry points the 8-GPU run depends on -- `init_process_group("nccl", device_id=...)`, the fused all-reduce
    of the domain sums, `all_to_all_single` with (empty) uneven splits on device buffers, and HIP-graph capture of a
    forward that contains them -- executed for real with backend nccl at world size 1 (`always_communicate=True`), in a
    child process so the process group cannot leak into other tests."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_worker, args=(port, q))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert all(res.values()), res


# ------------------------------------------------------------------------------------------------ partitioned TRAINING step
def _ref_loss(out, y, tm, cm, n):
    """main_graph_knowledge_transfer.py:44-54 on the whole graph (the form bench.py times)"""
    import torch.nn.functional as F
    lb, lt, lth = out[:3]
    tmt = tm & ~cm
    yi = y[:, None]
    nll = lambda logp, w: -(logp.gather(1, yi).squeeze(1) * w).sum()
    return (2 * nll(lb, tm.float() / tm.sum()) + nll(lt, tmt.float() / tmt.sum()) + nll(lth, tmt.float() / tmt.sum())) / 4 \
        + F.kl_div(lth, lt, log_target=True, reduction="batchmean")


def _train_worker(rank, world, port, q, layers):
    import copy
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bridged_gnn_amd import synth
        from bridged_gnn_amd.data import Data
        from bridged_gnn_amd.dist_train import PartitionedTrainer
        from bridged_gnn_amd.ktgnn import KTGNN_no_complement
        n = 6000
        ei, mask = synth.bridged_graph(3500, 2500, 4, 8, 7000, cluster=128, p_local=0.8, seed=4)
        torch.manual_seed(0)
        model = KTGNN_no_complement(64, 3, layers, 64, use_bn=True, dim_share=64, dropout=0.0).to(DEV).train()
        g = torch.Generator(device=DEV).manual_seed(1)
        x = torch.randn(n, 64, device=DEV, generator=g)
        y = torch.randint(0, 3, (n,), device=DEV, generator=g)
        tm = torch.rand(n, device=DEV, generator=g) < 0.5
        cm = _t(mask)
        data = Data(x=x, edge_index=_t(ei), central_mask=cm)
        ref = copy.deepcopy(model)
        tr = PartitionedTrainer(model, ei, mask, rank, world, DEV)
        own = tr.owned_global
        o_ref, o_par = torch.optim.SGD(ref.parameters(), lr=0.05), torch.optim.SGD(model.parameters(), lr=0.05)
        worst = {"loss": 0.0, "out": 0.0, "grad": 0.0, "param": 0.0, "bn": 0.0, "grad_of": ""}
        for step in range(3):
            o_ref.zero_grad(set_to_none=True)
            out_r = ref(data)
            loss_r = _ref_loss(out_r, y, tm, cm, n)
            loss_r.backward()
            o_par.zero_grad(set_to_none=True)
            out_p = tr.forward(x[own].contiguous())
            loss_p = tr.reference_loss(out_p, y[own], tm[own])
            loss_p.backward()
            tr.sync_grads()
            tot = loss_p.detach().double().cpu().reshape(1)
            dist.all_reduce(tot)
            worst["loss"] = max(worst["loss"], abs(float(tot) - float(loss_r)) / abs(float(loss_r)))
            for a, b in zip(out_p, out_r[:3]):
                worst["out"] = max(worst["out"], float((a - b[own]).abs().max()))
            # (a bias in front of a BatchNorm has an exactly-zero gradient: what both sides compute there is rounding noise, so the
            #  error of a tensor is taken relative to its own largest gradient plus 1e-3 of the largest gradient of the model)
            gmax = max(float(r.grad.abs().max()) for r in ref.parameters())
            for (nm, p), r in zip(model.named_parameters(), ref.parameters()):
                assert p.grad is not None and r.grad is not None, nm
                e = float((p.grad - r.grad).abs().max()) / (float(r.grad.abs().max()) + 1e-3 * gmax)
                if e > worst["grad"]:
                    worst["grad"], worst["grad_of"] = e, f"{nm} (|ref| max {float(r.grad.abs().max()):.2e}, model max {gmax:.2e})"
            o_ref.step(); o_par.step()
            for p, r in zip(model.parameters(), ref.parameters()):
                worst["param"] = max(worst["param"], float((p - r).abs().max()))
            for b1, b2 in zip(model.buffers(), ref.buffers()):
                if b1.dtype.is_floating_point:
                    worst["bn"] = max(worst["bn"], float((b1 - b2).abs().max()))
        q.put((rank, worst, tr.plan.summary()))
    except Exception:                                            # report instead of leaving the parent waiting for the queue
        import traceback
        q.put((rank, {"error": traceback.format_exc()}, {"n_halo": -1}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,layers", [(2, 2), (3, 2), (2, 3)])
def test_partitioned_training_step_matches_the_single_gpu_step(world, layers):
    """VERDICT r2 #7 / SURVEY 8(e)+(f1): three SGD steps of the reference loss with REAL ranks (gloo group, payload staged through
    the host because RCCL refuses two ranks per device; kernels are the production ones) against the same steps of the single-GPU
    training path on the whole graph: loss, owned outputs, ALL-REDUCED parameter gradients (sync BatchNorm statistics, reverse halo
    exchange of the gradient rows, the all-reduced adjoint of the domain means), parameters and BatchNorm buffers after each step.
    layers = 3: a hidden conv whose INPUT needs gradients (512-byte rows exchanged, gradient rows sent back)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q, layers)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, w, summ in res:
        print(rank, w, summ)
        assert "error" not in w, w["error"]
        assert summ["n_halo"] > 0
        assert w["loss"] < 2e-6 and w["out"] < 2e-5 and w["grad"] < 3e-3 and w["param"] < 2e-6 and w["bn"] < 1e-6, (rank, w)
        # (grad: worst tensor is the Linear in front of clf_transformer's BatchNorm, whose gradient is what survives the
        #  cancellation of the batch-mean terms -- 3.5e-6 absolute on a 2.5e-3 gradient between two fp32 evaluations with
        #  different reduction orders; every other tensor is below 1e-4)
