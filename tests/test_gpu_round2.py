"""GPU: parity cases added in round 2 -- BASELINE config 3 (Twitter stand-in, F=300 / hidden=128), layer_num in {1, 3},
the a4 glue (v2 encoders, PairNorm, get_probs_*), merge_graphs / reorder against reference-generated fixtures, the
full-size C4 forward with real attention on sampled rows, and the two advisor findings (CSR cache identity, eval-mode
backward through clf_transformer).  Float bar 1e-5 relative (tests/conftest.py:assert_close) unless a test says why not."""
import numpy as np
import pytest
import torch

from conftest import assert_close, sub
from oracle import oracle_c as OC
from oracle import oracle_np as O
from oracle import oracle_torch as OT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _load_model(sd, *ctor, **kw):
    from bridged_gnn_amd.ktgnn import KTGNN_no_complement
    model = KTGNN_no_complement(*ctor, **kw)
    model.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in sd.items()}, strict=True)
    return model.to(DEV).eval()


# ------------------------------------------------------------------------------------------------ config 3
def test_ktgnn_c3_twitter_standin_golden(golden):
    """BASELINE config 3: Twitter_Graph stand-in (581 S + 20 230 T, F=300, ~0.9 M random + kNN edges, undirected),
    2-layer KT-GNN hidden 128 (`run.sh:5-7`); expected outputs from the reference's own KTGNN.py (oracle/gen_golden.py).
    Din=300 is outside the W-stationary transform's envelope -> the tiled fp32-MFMA GEMM kernel runs here."""
    from bridged_gnn_amd import synth, utils
    from bridged_gnn_amd.data import Data
    k = golden("ktgnn_c3.npz")
    x, ei, y, m = synth.twitter_standin(seed=0)
    n = x.shape[0]
    assert synth.edge_hash(ei, n) == int(k["edge_hash"]), "stand-in graph differs from the one the fixture was made on"
    und = utils.to_undirected(_t(ei), n)
    assert und.shape[1] == int(k["n_edges_undirected"])
    model = _load_model(sub(k, "sd."), 300, 2, 2, 128, root_weight=False, use_bn=True, dim_share=300)
    data = Data(x=_t(x), edge_index=und, y=_t(y), central_mask=_t(m))
    with torch.no_grad():
        lb, lt, lth, _ = model(data)
        emb = model.get_emb(data)
    rows = k["rows"]
    assert_close(emb.cpu().numpy()[rows, :128], k["emb_rows"], what="get_emb")
    assert_close(lb.cpu().numpy()[rows], k["logp_base"], what="logp_base")
    assert_close(lt.cpu().numpy()[rows], k["logp_target"], what="logp_target")
    assert_close(lth.cpu().numpy()[rows], k["logp_target_hat"], what="logp_target_hat")
    sums = np.array([t.double().sum().item() for t in (lb, lt, lth)])
    assert np.allclose(sums, k["sums"], rtol=1e-5)
    # the HIP-graph replay of the same forward (what bench.py --config c3 times) returns the same tensors
    replay = model.graphed(data)
    g = replay()
    assert torch.allclose(g[0], lb, rtol=1e-6, atol=1e-6) and torch.allclose(g[2], lth, rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------------------------------------ layer_num
@pytest.mark.parametrize("tag,ctor,use_bn", [("l1", (32, 4, 1, 4), False), ("l3", (32, 3, 3, 32), True)])
def test_ktgnn_layer_num_1_and_3(golden, tag, ctor, use_bn):
    """KTGNN.py:344-358: the layer_num == 1 branch (one conv straight to `num_classes` columns feeding clf_*; hidden ==
    num_classes, no BatchNorm) and the three-layer chain, where a conv's column-sum epilogue feeds the next conv."""
    from bridged_gnn_amd import synth, utils
    from bridged_gnn_amd.data import Data
    k = golden("ktgnn_layers.npz")
    x, ei, y, m = synth.sync_rd_intra(n=1500, feat=32, homophily=0.7, deg=6, k_cross=8, seed=5)
    und = utils.to_undirected(_t(ei), 1500)
    model = _load_model(sub(k, f"{tag}.sd."), *ctor, root_weight=False, use_bn=use_bn, dim_share=32)
    assert len(model.convs) == {"l1": 1, "l3": 2}[tag]
    data = Data(x=_t(x), edge_index=und, y=_t(y), central_mask=_t(m))
    with torch.no_grad():
        lb, lt, lth, _ = model(data)
        emb = model.get_emb(data)
    assert_close(emb.cpu().numpy()[:, :k[f"{tag}.emb"].shape[1]], k[f"{tag}.emb"], what="get_emb")
    assert_close(lb.cpu().numpy(), k[f"{tag}.logp_base"], what="logp_base")
    assert_close(lt.cpu().numpy(), k[f"{tag}.logp_target"], what="logp_target")
    assert_close(lth.cpu().numpy(), k[f"{tag}.logp_target_hat"], what="logp_target_hat")


def test_layer_num_1_with_bn_fails_like_the_reference():
    """the reference indexes an empty `bns` list (KTGNN.py:425) when layer_num == 1 and use_bn: IndexError there and here"""
    from bridged_gnn_amd import synth
    from bridged_gnn_amd.data import Data
    from bridged_gnn_amd.ktgnn import KTGNN_no_complement
    ei, mask = synth.random_multigraph(100, 500, seed=1)
    model = KTGNN_no_complement(8, 4, 1, 4, use_bn=True, dim_share=8).to(DEV).eval()
    with pytest.raises(IndexError), torch.no_grad():
        model(Data(x=torch.randn(100, 8, device=DEV), edge_index=_t(ei), central_mask=_t(mask)))


# ------------------------------------------------------------------------------------------------ a4 glue
def test_a4_encoders_pairnorm_and_get_probs_vs_reference(golden):
    """models/models.py:29-64 (PairNorm), :880-893 (MLP.forward), :1092-1096 (encode), :1122-1142 (get_probs_*): the
    BridgeScorer built from the shipped office A->D checkpoint against the reference's outputs on the shipped features."""
    from bridged_gnn_amd.bridge import BridgeScorer
    from bridged_gnn_amd.data import Data
    f, g, kf = golden("a4_office_a2d.npz"), golden("office_a2d_graph.npz"), golden("knn_office_a2d.npz")
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sub(f, "sd.").items()}
    sd.update({"source_learner.sim_net." + k: torch.from_numpy(np.asarray(v)) for k, v in sub(kf, "sim.").items()})
    ns = int(kf["n_src"])
    ds, dt = Data(x=_t(g["x"][:ns])), Data(x=_t(g["x"][ns:]))
    sc = BridgeScorer(sd, DEV)
    assert sc.version == "v2" and sc.sim_mode == "mlp"
    zs, zt = sc.encode_source(ds), sc.encode_target(dt)
    assert_close(zs.cpu().numpy(), kf["z_src"], what="encode_source")
    assert_close(zt.cpu().numpy(), kf["z_tar"], what="encode_target")
    for mode in ("PN", "PN-SI", "PN-SCS"):
        scm = BridgeScorer(sd, DEV, norm_mode=mode)
        # PairNorm divides by a norm of O(10..100)-term fp32 sums whose order differs between torch-CPU and the GPU
        assert_close(scm.encode_source(ds).cpu().numpy()[::8], f[f"z_src_{mode}"], what=f"z_src {mode}")
        assert_close(scm.encode_target(dt).cpu().numpy()[::2], f[f"z_tar_{mode}"], what=f"z_tar {mode}")
    i1, i2 = _t(f["cross_idx1"]), _t(f["cross_idx2"])
    p, pcs, pct, z1, z2 = sc.get_probs_cross_domain(ds, dt, i1, i2, return_representation=True)
    assert p.shape == (600, 1) and torch.equal(z1, zs) and torch.equal(z2, zt)
    assert_close(p.cpu().numpy(), f["cross_probs"], what="get_probs_cross_domain")
    assert_close(pcs.cpu().numpy(), f["probs_clf_src"], what="probs_clf_src")
    assert_close(pct.cpu().numpy(), f["probs_clf_tar"], what="probs_clf_tar")
    p, pc = sc.get_probs_within_domain(ds, _t(f["src_idx1"]), _t(f["src_idx2"]), domain="source")
    assert_close(p.cpu().numpy(), f["src_probs"], what="get_probs_within_domain(source)")
    assert_close(pc.cpu().numpy(), f["src_probs_clf"], what="within source clf probs")
    p, pc = sc.get_probs_within_domain(dt, _t(f["tar_idx1"]), _t(f["tar_idx2"]), domain="target")
    assert_close(p.cpu().numpy(), f["tar_probs"], what="get_probs_within_domain(target)")
    assert_close(pc.cpu().numpy(), f["tar_probs_clf"], what="within target clf probs")
    # the pair entry point and the streaming top-k kernel agree on the winners' probabilities
    idx, probs, _ = sc.topk(zs, zt[:64], 20)
    q = torch.arange(64, device=DEV).repeat_interleave(20)
    again = sc.pair_probs(zs, zt[:64], idx.reshape(-1), q)
    assert_close(again.cpu().numpy(), probs.reshape(-1).cpu().numpy(), what="pair_probs vs top-k values")


# ------------------------------------------------------------------------------------------------ a9 / f4
def test_merge_graphs_and_reorder_vs_reference_fixture(golden):
    """main_bridged_graph.py:163-193 and :195-222 on the GPU against the reference's own outputs (bit-exact)."""
    from bridged_gnn_amd import bridge
    from bridged_gnn_amd.data import Data
    a, kf, g = golden("assembly_office_a2d.npz"), golden("knn_office_a2d.npz"), golden("office_a2d_graph.npz")
    ns = int(kf["n_src"])
    x, y = g["x"], g["y"]
    i64 = lambda v: _t(v.astype(np.int64))
    ds = Data(x=_t(x[:ns]), edge_index=i64(a["ei_src"]), y=_t(y[:ns]), train_mask=_t(g["train_mask"][:ns]))
    dt = Data(x=_t(x[ns:]), edge_index=i64(a["ei_tar"]), y=_t(y[ns:]), train_mask=_t(g["train_mask"][ns:]),
              val_mask=_t(g["val_mask"][ns:]), test_mask=_t(g["test_mask"][ns:]))
    ec = i64(kf["cross_edge_index"])
    ec_before = ec.clone()
    merged = bridge.merge_graphs(ds, dt, ec, i64(kf["within_src_edge_index"]), i64(kf["within_tar_edge_index"]))
    assert torch.equal(ec, ec_before), "the cross-edge argument must not be modified (the reference mutates it, :170)"
    assert np.array_equal(merged.edge_index.cpu().numpy(), a["merged_edge_index"])
    for key, ref in (("y", "merged_y"), ("train_mask", "merged_train"), ("val_mask", "merged_val"), ("test_mask", "merged_test"),
                     ("central_mask", "merged_central")):
        assert np.array_equal(getattr(merged, key).cpu().numpy(), a[ref]), key
    assert np.array_equal(merged.x[:, 0].cpu().numpy(), a["merged_x_col0"])
    m_src = {int(o): i for i, o in enumerate(a["orig_src"])}              # the reference's dict form (utils.py:58-63)
    ro = bridge.reorder(merged, ds, m_src, torch.from_numpy(a["orig_tar"].astype(np.int64)))   # and the tensor form
    assert np.array_equal(ro.edge_index.cpu().numpy(), a["reordered_edge_index"])
    for key, ref in (("y", "reordered_y"), ("train_mask", "reordered_train"), ("val_mask", "reordered_val"),
                     ("test_mask", "reordered_test"), ("central_mask", "reordered_central")):
        assert np.array_equal(getattr(ro, key).cpu().numpy(), a[ref]), key
    assert np.array_equal(ro.x[:, 0].cpu().numpy(), a["reordered_x_col0"])


# ------------------------------------------------------------------------------------------------ advisor findings
def test_csr_cache_is_keyed_by_tensor_identity_and_version():
    """the reference-parity call `conv(x, edge_index, e1, e2, mask)` caches its CSR: a second edge_index of the same
    shape that the caching allocator puts on the freed block, or an in-place edit, must not reuse the stale CSR."""
    from bridged_gnn_amd import ops
    from bridged_gnn_amd.ktgnn import AdaptedConv
    n, E = 500, 4000
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    conv = AdaptedConv(16, 8, root_weight=False).to(DEV).eval()
    x = torch.randn(n, 16, device=DEV)
    mask = _t(rng.random(n) < 0.5)

    def expect(ei):
        with torch.no_grad():
            return conv(x, None, central_mask=mask, csr=ops.build_dst_csr(ei, n, rewrite_self_loops=False))
    ptrs = []
    for trial in range(3):
        ei = _t(rng.integers(0, n, (2, E)))
        ptrs.append(ei.data_ptr())
        with torch.no_grad():
            got = conv(x, ei, None, None, mask)
        assert torch.equal(got, expect(ei)), f"stale CSR on trial {trial}"
        del ei                                         # the next tensor of the same size lands on the same block
    with torch.no_grad():
        ei = _t(rng.integers(0, n, (2, E)))
        a = conv(x, ei, None, None, mask)
        ei[0, :100] = (ei[0, :100] + 1) % n            # in-place edit: same object, same pointer, new version
        b = conv(x, ei, None, None, mask)
    assert torch.equal(b, expect(ei)) and not torch.equal(a, b)


def test_eval_mode_backward_reaches_clf_transformer():
    """model.eval() with grad enabled (frozen-BN fine-tuning, input attribution): every parameter -- the two Linears of
    clf_transformer included -- and the input get the gradients of the CPU fp64 autograd oracle."""
    from bridged_gnn_amd import synth
    from bridged_gnn_amd.data import Data
    from bridged_gnn_amd.ktgnn import KTGNN_no_complement
    n, feat, hidden, C = 500, 12, 64, 3              # hidden 64: inside the raw W-stationary kernel's envelope
    ei, mask = synth.random_multigraph(n, 4000, frac_src=0.5, seed=5)
    rng = np.random.default_rng(6)
    x = rng.standard_normal((n, feat)).astype(np.float32)
    w = [rng.standard_normal((n, C)) for _ in range(3)]
    torch.manual_seed(4)
    model = KTGNN_no_complement(feat, C, 2, hidden, use_bn=True, dim_share=feat)
    g = torch.Generator().manual_seed(2)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).eval()
    xg = _t(x).requires_grad_(True)
    out = model(Data(x=xg, edge_index=_t(ei), central_mask=_t(mask)))
    sum((o * _t(wi).float()).sum() for o, wi in zip(out[:3], w)).backward()
    # oracle: the reference's op sequence in fp64 on the CPU
    ref = KTGNN_no_complement(feat, C, 2, hidden, use_bn=True, dim_share=feat)
    ref.load_state_dict(sd0)
    ref = ref.double().eval()
    mo = torch.from_numpy(mask)
    e1, e2 = OT.graph_partition(torch.from_numpy(ei), mo)
    xo = torch.from_numpy(x).double().requires_grad_(True)
    conv = lambda c, xx: OT.adaptedconv(xx, mo, e1, e2, dict(c.named_parameters()))
    h = torch.relu(ref.bns[0](conv(ref.convs[0], xo)))
    outs = (torch.log_softmax(conv(ref.clf_base, h), 1), torch.log_softmax(conv(ref.clf_target, h), 1),
            torch.log_softmax(conv(ref.clf_target, ref.clf_transformer(h)), 1))
    sum((o * torch.from_numpy(wi)).sum() for o, wi in zip(outs, w)).backward()
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    for o, r in zip(out[:3], outs):
        assert rel(o.detach().cpu().double(), r.detach()) < 2e-5
    assert rel(xg.grad.cpu().double(), xo.grad) < 2e-4, "dL/dx"
    refp = dict(ref.named_parameters())
    for name, prm in model.named_parameters():
        assert prm.grad is not None, f"{name} got no gradient"
        assert rel(prm.grad.cpu().double(), refp[name].grad) < 2e-4, name


# ------------------------------------------------------------------------------------------------ C4 at full size
def test_c4_full_size_forward_real_attention_vs_oracle():
    """The bench's own model (seeded weights, non-trivial BatchNorm) on the bench's own 1M-node / 21M-edge graph, real
    attention: the hidden conv output (BN + ReLU epilogue on) and the three log-prob heads against a full forward of the
    C oracle -- strictly (1e-5) on >= 512 sampled destination rows of both domains including the max-degree rows, and
    over ALL rows as well."""
    import argparse
    import bench
    from bridged_gnn_amd.data import Data
    args = argparse.Namespace(config="c4", nodes=1_000_000, edges=20_000_000, graph="local", feat=128, hidden=128, classes=2)
    wl = bench.make_workload(args, torch.device(DEV))
    model = bench.build_model(args, DEV)
    ei, mask = wl["ei_np"], wl["mask_np"]
    data = Data(x=wl["x"], edge_index=_t(ei), central_mask=_t(mask))
    with torch.no_grad():
        out = [t.cpu().numpy() for t in model(data)[:3]]
        emb = model.get_emb(data).cpu().numpy()[:, :128]
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    rowptr, col, _ = O.dst_csr(ei, mask)
    assert int(rowptr[-1]) == 20_991_058
    rb, rt, rth, remb = OC.ktgnn_forward_eval(wl["x"].cpu().numpy(), rowptr, col, mask, sd, return_emb=True)
    deg = np.diff(rowptr)
    n = mask.shape[0]
    rows = np.unique(np.concatenate([np.arange(0, n, 1999), np.argsort(deg[: n // 2])[-64:], n // 2 + np.argsort(deg[n // 2:])[-64:],
                                     np.argsort(deg)[:16]]))
    assert rows.shape[0] >= 512 and mask[rows].sum() >= 200 and (~mask[rows]).sum() >= 200
    for name, got, ref in (("hidden conv (BN+ReLU)", emb, remb), ("logp_base", out[0], rb), ("logp_target", out[1], rt),
                           ("logp_target_hat", out[2], rth)):
        assert_close(got[rows], ref[rows], what=f"{name}, sampled rows")
        assert_close(got, ref, what=f"{name}, all rows")
    # the checksums bench.py prints for the timed forward are those of this output
    sums = [float(np.asarray(o, np.float64).sum()) for o in out]
    assert np.allclose(sums, [float(np.asarray(r, np.float64).sum()) for r in (rb, rt, rth)], rtol=1e-6)


def test_c4_full_size_aggregation_backward_vs_torch_autograd():
    """Row f1 at the size the bench times: the hidden conv's aggregation (D = 128, pull-form backward, hub segments where the
    graph has them) and the three D = 2 classifier heads in one walk (log_softmax epilogue), forward + backward on the bench's
    own 1M-node / 21M-edge graph, against torch autograd of the reference's op sequence (index_select / leaky_relu / segment
    softmax / index_add, KTGNN.py:292-305) on the SAME fp32 tables on the GPU (so no leaky-relu kink can flip between the
    sides).  Bars: 5e-6 of the tensor's max for outputs and table gradients (measured <= 6e-7), 1e-4 for the attention vectors'
    gradients (sums over all 21M edges, fp32 atomics in an unordered sequence on BOTH sides; measured 7e-6 .. 3e-5)."""
    import argparse
    import torch.nn.functional as F
    import bench
    from bridged_gnn_amd import ops
    from bridged_gnn_amd.ktgnn import _AggregateFn, _AggregateHeadsFn
    args = argparse.Namespace(config="c4", nodes=1_000_000, edges=20_000_000, graph="local", feat=128, hidden=128, classes=2)
    wl = bench.make_workload(args, torch.device(DEV))
    ei, mask = wl["ei_np"], wl["mask_np"]
    n = mask.shape[0]
    csr = ops.build_dst_csr(_t(ei), n)
    assert csr.num_edges == 20_991_058
    m8 = _t(mask).to(torch.uint8)
    e1, e2 = (e.to(DEV) for e in OT.graph_partition(torch.from_numpy(ei), torch.from_numpy(mask)))
    dst = torch.cat((e1[1], e2[1]))
    n1 = e1.shape[1]

    def seg_softmax(z):
        m = torch.full((n,), float("-inf"), device=DEV).scatter_reduce(0, dst, z, reduce="amax", include_self=True)
        e = (z - m[dst]).exp()
        return e / (torch.zeros(n, device=DEV).index_add_(0, dst, e)[dst] + 1e-16)

    def ref_conv(t1, t2, b1, b2):
        z = torch.cat((F.leaky_relu(t1[e1[0]] + t1[e1[1]], 0.1) @ b1, F.leaky_relu(t2[e2[0]] + t2[e2[1]], 0.1) @ b2))
        al = seg_softmax(z)
        o = torch.zeros(n, t1.shape[1], device=DEV)
        return o.index_add(0, e1[1], t1[e1[0]] * al[:n1, None]).index_add(0, e2[1], t2[e2[0]] * al[n1:, None])

    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    g = torch.Generator(device=DEV).manual_seed(11)
    # ---- hidden conv: D = 128
    D = 128
    T = [torch.randn(n, D, device=DEV, generator=g) for _ in range(2)]
    A = [torch.randn(D, device=DEV, generator=g) * 0.1 for _ in range(2)]
    w = torch.randn(n, D, device=DEV, generator=g)
    mine = [t.clone().requires_grad_(True) for t in T + A]
    out = _AggregateFn.apply(mine[0], mine[1], mine[2], mine[3], csr, m8, D, 0.1)
    (out[:, :D] * w).sum().backward()
    theirs = [t.clone().requires_grad_(True) for t in T + A]
    o = ref_conv(*theirs)
    (o * w).sum().backward()
    errs = {"out": rel(out[:, :D].detach(), o.detach())}
    for name, a, b in zip(("dh_t2s", "dh_s2t", "da_t2s", "da_s2t"), mine, theirs):
        errs[name] = rel(a.grad, b.grad)
    print("C4 full size, D=128:", {k: f"{v:.1e}" for k, v in errs.items()})
    assert max(errs["out"], errs["dh_t2s"], errs["dh_s2t"]) < 5e-6, errs       # measured 2.4e-7 .. 3.8e-7
    assert max(errs["da_t2s"], errs["da_s2t"]) < 1e-4, errs
    del out, o, mine, theirs, T, w
    torch.cuda.empty_cache()
    # ---- the three classifier heads: D = 2, tables [N, 4] per head, log_softmax in the epilogue (KTGNN.py:432-435)
    D, H = 2, 3
    tabs = []
    for _ in range(2 * H):
        t = torch.zeros(n, 4, device=DEV)
        t[:, :D] = torch.randn(n, D, device=DEV, generator=g)
        tabs.append(t)
    a_t = torch.randn(H, D, device=DEV, generator=g) * 0.5
    a_s = torch.randn(H, D, device=DEV, generator=g) * 0.5
    w = torch.randn(n, H, D, device=DEV, generator=g)
    mine = [t.clone().requires_grad_(True) for t in [a_t, a_s] + tabs]
    lp = _AggregateHeadsFn.apply(csr, m8, D, 0.1, *mine)                     # [N, H, 4]
    (lp[:, :, :D] * w).sum().backward()
    theirs = [t.clone().requires_grad_(True) for t in [a_t, a_s] + tabs]
    ref = torch.stack([torch.log_softmax(ref_conv(theirs[2 + 2 * h][:, :D], theirs[3 + 2 * h][:, :D], theirs[0][h], theirs[1][h]), 1)
                       for h in range(H)], dim=1)                            # [N, H, D]
    (ref * w).sum().backward()
    errs = {"logp": rel(lp[:, :, :D].detach(), ref.detach()), "da_t2s": rel(mine[0].grad, theirs[0].grad), "da_s2t": rel(mine[1].grad, theirs[1].grad)}
    for i in range(2 * H):
        errs[f"dtab{i}"] = rel(mine[2 + i].grad[:, :D], theirs[2 + i].grad[:, :D])
        assert float(mine[2 + i].grad[:, D:].abs().max()) == 0.0             # padding columns get no gradient
    print("C4 full size, 3 heads D=2:", {k: f"{v:.1e}" for k, v in errs.items()})
    assert max(v for k, v in errs.items() if not k.startswith("da_")) < 5e-6, errs      # measured 1.6e-7 .. 5.7e-7
    assert max(errs["da_t2s"], errs["da_s2t"]) < 1e-4, errs


def _small_training_setup(drop):
    import copy
    from bridged_gnn_amd import synth, utils
    from bridged_gnn_amd.data import Data
    from bridged_gnn_amd.ktgnn import KTGNN_no_complement
    x, ei, y, m = synth.sync_rd_intra(n=4000, feat=64, homophily=0.7, deg=8, k_cross=10, seed=2)
    und = utils.to_undirected(torch.from_numpy(ei).to(DEV), x.shape[0])
    data = Data(x=torch.from_numpy(x).to(DEV), edge_index=und, central_mask=torch.from_numpy(m).to(DEV))
    yt = torch.from_numpy(y).to(DEV).clamp_min(0)[:, None]
    g = torch.Generator(device=DEV).manual_seed(5)
    w = torch.rand(x.shape[0], device=DEV, generator=g); w /= w.sum()
    loss_fn = lambda out: -sum((o.gather(1, yt).squeeze(1) * w).sum() for o in out[:3])
    torch.manual_seed(0)
    a = KTGNN_no_complement(64, 2, 2, 64, use_bn=True, dim_share=64, dropout=drop).to(DEV).train()
    return data, loss_fn, a, copy.deepcopy(a)


@pytest.mark.parametrize("opt_name", ["sgd", "adam"])
def test_graphed_train_step_follows_the_eager_loop(opt_name):
    """`KTGNN_no_complement.graphed_train_step` (zero_grad + forward + loss + backward + optimizer.step of
    main_graph_knowledge_transfer.py:39-68 as ONE HIP graph) against the same loop run eagerly on a twin model, the two
    interleaved step by step (eager work between replays is what exposed hipMemsetAsync nodes replaying a stale pattern):
    losses agree, parameters agree, and an eager eval forward afterwards sees the trained weights (no stale packed copies)."""
    from bridged_gnn_amd.ktgnn import KTGNN_no_complement
    data, loss_fn, a, b = _small_training_setup(0.0)
    mk = (lambda ps: torch.optim.SGD(ps, lr=0.05)) if opt_name == "sgd" else (lambda ps: torch.optim.Adam(ps, lr=1e-3, capturable=True))
    oa, ob = mk(a.parameters()), mk(b.parameters())
    W = 2
    run = a.graphed_train_step(data, loss_fn, oa, warmup=W)

    def eager():
        ob.zero_grad(set_to_none=True)
        l = loss_fn(b(data)); l.backward(); ob.step()
        return float(l.detach())
    for _ in range(W):
        eager()
    for i in range(25):
        la, lb = float(run().detach()), eager()
        # SGD: the two runs differ by the order of fp32 atomics only.  Adam normalises every coordinate's step to ~lr, also the ones whose
        # gradient is rounding noise (exactly-zero gradients before a BatchNorm), so two runs of the SAME loop drift apart slowly
        tol = 2e-5 if opt_name == "sgd" else (1e-3 if i < 8 else 2e-2)
        assert abs(la - lb) <= tol * abs(lb), (i, la, lb)
    if opt_name == "sgd":                      # (Adam turns the rounding noise of the exactly-zero gradients before a BatchNorm into +-lr steps)
        for (n_, p), q in zip(a.named_parameters(), b.parameters()):
            assert float((p.detach() - q.detach()).abs().max()) <= 1e-4 * float(q.detach().abs().max()) + 1e-7, n_
    a.eval()
    fresh = KTGNN_no_complement(64, 2, 2, 64, use_bn=True, dim_share=64).to(DEV).eval()
    fresh.load_state_dict(a.state_dict())
    with torch.no_grad():
        for u, v in zip(a(data)[:3], fresh(data)[:3]):
            assert torch.allclose(u, v, rtol=1e-6, atol=1e-6)
    with pytest.raises(RuntimeError):
        a.graphed_train_step(data, loss_fn, oa)                    # eval mode
    a.train()
    with pytest.raises(RuntimeError):
        a.graphed_train_step(data, loss_fn, torch.optim.Adam(a.parameters(), lr=1e-3))    # not capturable


def test_eval_passes_between_graph_replays_see_fresh_batchnorm_state():
    """Regression (round-2 advisor, high): `bn_eval_affine` caches BatchNorm(eval) scale / shift per host-side version
    counters, which a replayed HIP graph never advances -- from the SECOND eval pass on the hidden BN used the parameters
    and running statistics frozen at the first one.  Interleave eval forwards with replays (the workflow of
    tools/train_300_epochs.py) and compare every one of them with a fresh model loaded from the state_dict."""
    from bridged_gnn_amd.ktgnn import KTGNN_no_complement
    data, loss_fn, a, _ = _small_training_setup(0.5)
    run = a.graphed_train_step(data, loss_fn, torch.optim.SGD(a.parameters(), lr=0.05), warmup=2)
    prev = None
    for rnd in range(4):
        for _ in range(3):
            run()
        a.eval()
        fresh = KTGNN_no_complement(64, 2, 2, 64, use_bn=True, dim_share=64).to(DEV).eval()
        fresh.load_state_dict(a.state_dict())
        with torch.no_grad():
            got, want = [o.clone() for o in a(data)[:3]], fresh(data)[:3]
        for u, v in zip(got, want):
            assert torch.allclose(u, v, rtol=1e-6, atol=1e-6), rnd
        if prev is not None:                                      # the weights did move between the eval passes
            assert not torch.allclose(prev[0], got[0], rtol=1e-4, atol=1e-5)
        prev = got
        a.train()


def test_eager_training_with_frozen_bn_affine_keeps_eval_caches_fresh():
    """Regression (same finding): the fused BN+ReLU+dropout kernel writes running_mean / running_var through raw pointers, so
    with the BN affine parameters frozen nothing used to advance a version counter and the eval-mode caches
    (`bn_eval_affine`, `_fold_transformer`) kept the statistics of the first eval pass."""
    from bridged_gnn_amd.ktgnn import KTGNN_no_complement
    data, loss_fn, a, _ = _small_training_setup(0.0)
    for bn in list(a.bns) + [a.clf_transformer[1]]:
        bn.weight.requires_grad_(False); bn.bias.requires_grad_(False)
    opt = torch.optim.SGD([p for p in a.parameters() if p.requires_grad], lr=0.0)    # lr 0: ONLY the running statistics move
    for rnd in range(3):
        a.eval()
        fresh = KTGNN_no_complement(64, 2, 2, 64, use_bn=True, dim_share=64).to(DEV).eval()
        fresh.load_state_dict(a.state_dict())
        with torch.no_grad():
            for u, v in zip(a(data)[:3], fresh(data)[:3]):
                assert torch.allclose(u, v, rtol=1e-6, atol=1e-6), rnd
        a.train()
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            loss_fn(a(data)).backward()
            opt.step()


def test_graphed_train_step_draws_a_new_dropout_mask_per_replay():
    """the seed baked into the graph + the device step counter advanced inside it: with the weights held still (lr = 0) every
    replay still sees another dropout mask, i.e. another loss"""
    data, loss_fn, a, _ = _small_training_setup(0.5)
    run = a.graphed_train_step(data, loss_fn, torch.optim.SGD(a.parameters(), lr=0.0))
    losses = [float(run().detach()) for _ in range(6)]
    assert len(set(losses)) == len(losses), losses
    assert int(run.step.item()) >= 6


def test_graph_replays_survive_eager_work_between_them():
    """Regression: the aggregation's tile counters used to be cleared with hipMemsetAsync; captured after a warm-up on a side stream,
    that memset NODE filled them with a stale pattern as soon as eager work had run between two replays, and the replay computed
    nothing (bgnn_zero_async in csrc/bgnn_common.h is a kernel of our own now)."""
    from bridged_gnn_amd import ops, synth
    n, D = 6000, 128
    ei, mask = synth.random_multigraph(n, 10 * n, frac_src=0.4, n_isolated=2, seed=9)
    csr = ops.build_dst_csr(_t(ei), n)
    m8 = _t(mask).to(torch.uint8)
    g = torch.Generator(device=DEV).manual_seed(1)
    h1, h2 = torch.randn(n, D, device=DEV, generator=g), torch.randn(n, D, device=DEV, generator=g)
    a1, a2 = torch.randn(D, device=DEV, generator=g) * 0.1, torch.randn(D, device=DEV, generator=g) * 0.1
    f = lambda A, B: ops.adaptedconv_aggregate(A, B, a1, a2, csr, m8, D, 0.1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            f(h1, h2)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        h1.mul_(0.97); h2.mul_(0.97)
        res = f(h1, h2)
    for i in range(5):
        gr.replay()
        torch.cuda.synchronize()
        ref = f(h1.clone(), h2.clone())                            # eager work between the replays (and the expected value)
        assert torch.equal(res, ref), i


def test_total_sum_and_column_sums():
    from bridged_gnn_amd import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(70001, device=DEV, generator=g).requires_grad_(True)
    s = ops.total_sum(x)
    assert abs(float(s) - float(x.detach().double().sum())) < 1e-3
    s.backward()
    assert torch.equal(x.grad, torch.ones_like(x))
    y = torch.randn(5000, 128, device=DEV, generator=g)
    assert torch.allclose(ops.column_sums(y), y.double().sum(0).float(), rtol=1e-6, atol=1e-5)


def test_input_domain_sums_cache_follows_x():
    """the cached domain sums of the (static) input features are dropped when x is written in place or replaced"""
    from bridged_gnn_amd import synth
    from bridged_gnn_amd.data import Data
    from bridged_gnn_amd.ktgnn import KTGNN_no_complement
    ei, mask = synth.random_multigraph(800, 6000, seed=3)
    torch.manual_seed(0)
    model = KTGNN_no_complement(32, 3, 2, 64, use_bn=True, dim_share=32).to(DEV).eval()
    x = torch.randn(800, 32, device=DEV)
    data = Data(x=x, edge_index=_t(ei), central_mask=_t(mask))
    with torch.no_grad():
        a = [t.clone() for t in model(data)[:3]]
        b = model(data)[:3]                                   # second call: sums come from the cache
        close = lambda u, v: torch.allclose(u, v, rtol=1e-6, atol=1e-6)   # (the fp64 column-sum atomics are not ordered run to run)
        assert all(close(u, v) for u, v in zip(a, b))
        x.mul_(1.5)                                           # in-place edit: new version
        c = [t.clone() for t in model(data)[:3]]
        fresh = KTGNN_no_complement(32, 3, 2, 64, use_bn=True, dim_share=32).to(DEV).eval()
        fresh.load_state_dict(model.state_dict())
        d = fresh(Data(x=x.clone(), edge_index=_t(ei), central_mask=_t(mask)))[:3]
        assert all(close(u, v) for u, v in zip(c, d)) and not close(a[0], c[0])


def _hub_graph(n, n_hubs, hub_deg, seed):
    """random multigraph + `n_hubs` destination rows with ~hub_deg in-edges each (both domains)"""
    from bridged_gnn_amd import synth
    ei, mask = synth.random_multigraph(n, 6 * n, frac_src=0.4, n_isolated=2, seed=seed)
    rng = np.random.default_rng(seed)
    hubs = rng.choice(n, size=n_hubs, replace=False)
    extra = np.stack([rng.integers(0, n, size=n_hubs * hub_deg), np.repeat(hubs, hub_deg)])
    return np.concatenate([ei, extra], axis=1).astype(np.int64), mask, hubs


@pytest.mark.parametrize("D,heads", [(128, 1), (64, 1), (2, 3), (4, 2)])
def test_hub_rows_segments_equal_the_single_walk(D, heads, monkeypatch):
    """Rows above ops.HUB_THRESHOLD are cut into segments, parked and merged (bgnn_adaptedconv_aggregate_hub_f32): same result
    as the plain launch (1e-6 of the row scale; the summation order inside a hub row differs), epilogue, column sums and
    the part-3 softmax state included, and vs the C oracle."""
    from bridged_gnn_amd import ops
    from oracle import oracle_c as OC
    n = 4000
    ei, mask, hubs = _hub_graph(n, 7, 900, seed=D + heads)
    csr = ops.build_dst_csr(torch.from_numpy(ei).to(DEV), n)
    tabs = csr.hub_tables()
    assert tabs is not None and set(hubs.tolist()) <= set(tabs[0].cpu().tolist())
    deg = (csr.rowptr[1:] - csr.rowptr[:-1]).cpu().numpy()
    assert int(tabs[3].numel()) == int(sum((deg[h] + ops.HUB_SEGMENT - 1) // ops.HUB_SEGMENT for h in tabs[0].cpu().tolist()))
    g = torch.Generator(device=DEV).manual_seed(D)
    ld = ops.pad4(D)
    m8 = torch.from_numpy(mask).to(DEV).to(torch.uint8)
    t1 = torch.zeros(n, heads * ld, device=DEV); t2 = torch.zeros(n, heads * ld, device=DEV)
    for h in range(heads):
        t1[:, h * ld:h * ld + D] = torch.randn(n, D, device=DEV, generator=g)
        t2[:, h * ld:h * ld + D] = torch.randn(n, D, device=DEV, generator=g)
    a1 = torch.randn(heads, D, device=DEV, generator=g) * 0.3
    a2 = torch.randn(heads, D, device=DEV, generator=g) * 0.3
    kw = {}
    if heads == 1:
        a1, a2 = a1[0].contiguous(), a2[0].contiguous()
        kw = dict(ep_scale=torch.rand(D, device=DEV, generator=g) + 0.5, ep_shift=torch.randn(D, device=DEV, generator=g) * 0.1, ep_relu=True)

    def run():
        extra = dict(kw)
        ms = None
        if heads == 1:
            extra["colsum"] = torch.zeros(2 * ld + 2, dtype=torch.float64, device=DEV)
        else:
            ms = torch.zeros(n, heads, 2, device=DEV)
            extra.update(log_softmax=True, state_ms=ms, part=3)
        out = ops.adaptedconv_aggregate(t1, t2, a1, a2, csr, m8, D, 0.1, heads=heads, **extra)
        return out, extra.get("colsum"), ms
    out_h, cs_h, ms_h = run()
    monkeypatch.setenv("BGNN_HUB_ROWS", "0")
    out_p, cs_p, ms_p = run()
    monkeypatch.delenv("BGNN_HUB_ROWS")
    scale = float(out_p.abs().max())
    assert float((out_h - out_p).abs().max()) <= 2e-6 * max(scale, 1.0)
    if cs_h is not None:
        assert torch.allclose(cs_h, cs_p, rtol=1e-6, atol=1e-5)
        assert torch.equal(cs_h[-2:], cs_p[-2:])
    if ms_h is not None:
        # the softmax state of a hub row: same max, sum within rounding
        assert torch.equal(ms_h[:, :, 0], ms_p[:, :, 0]) and torch.allclose(ms_h[:, :, 1], ms_p[:, :, 1], rtol=1e-5)
    if heads == 1:                               # the training forward: attention coefficients of the hub rows' edges too
        o_h, al_h = ops.adaptedconv_aggregate(t1, t2, a1, a2, csr, m8, D, 0.1, want_alpha=True)
        monkeypatch.setenv("BGNN_HUB_ROWS", "0")
        o_p, al_p = ops.adaptedconv_aggregate(t1, t2, a1, a2, csr, m8, D, 0.1, want_alpha=True)
        monkeypatch.delenv("BGNN_HUB_ROWS")
        assert torch.allclose(al_h, al_p, rtol=2e-5, atol=1e-9) and float((o_h - o_p).abs().max()) <= 2e-6 * max(float(o_p.abs().max()), 1.0)
    if heads == 1:                               # vs the C oracle (fp64 truth of the same fp32 tables), without epilogue
        out_raw = ops.adaptedconv_aggregate(t1, t2, a1, a2, csr, m8, D, 0.1)
        ref = OC.adaptedconv_aggregate_f64(t1[:, :D].contiguous().cpu().numpy(), t2[:, :D].contiguous().cpu().numpy(), a1.cpu().numpy(),
                                           a2.cpu().numpy(), csr.rowptr.cpu().numpy(), csr.col.cpu().numpy(), mask, 0.1)
        assert np.abs(out_raw[:, :D].cpu().numpy() - ref).max() <= 2e-6 * max(np.abs(ref).max(), 1.0)


def test_hub_rows_in_the_model_forward_c3_like():
    """the eval forward and the training step on a graph with hub rows (the Twitter stand-in's 581 source nodes have ~750 in-edges):
    hub path == plain path within 1e-5, gradients flow (the training aggregation of the three heads takes the hub path too)."""
    import os
    from bridged_gnn_amd import synth, utils
    from bridged_gnn_amd.data import Data
    from bridged_gnn_amd.ktgnn import KTGNN_no_complement
    x, ei, y, m = synth.twitter_standin(seed=0)
    und = utils.to_undirected(torch.from_numpy(ei).to(DEV), x.shape[0])
    torch.manual_seed(0)
    model = KTGNN_no_complement(300, 2, 2, 128, use_bn=True, dim_share=300).to(DEV).eval()
    data = Data(x=torch.from_numpy(x).to(DEV), edge_index=und, central_mask=torch.from_numpy(m).to(DEV))
    with torch.no_grad():
        a = [t.clone() for t in model(data)[:3]]
        assert model._csr.hub_tables() is not None
        os.environ["BGNN_HUB_ROWS"] = "0"
        try:
            b = [t.clone() for t in model(data)[:3]]
        finally:
            del os.environ["BGNN_HUB_ROWS"]
    for u, v in zip(a, b):
        assert torch.allclose(u, v, rtol=1e-5, atol=1e-5)
    model.train()
    lb, lt, lth, _ = model(data)
    (lb.sum() + lt.sum() + lth.sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
