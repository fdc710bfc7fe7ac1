"""ORACLE / TEST INFRASTRUCTURE -- build-container only.  Generates tests/golden/*.npz.

Runs the REFERENCE's own code (imported read-only from /root/reference under oracle/shim) on small
seeded inputs and on the reference's shipped artefacts, and stores inputs + expected outputs as
plain arrays.  Only arrays are committed; no reference source/bytecode travels.
Re-run:  python oracle/gen_golden.py        (deterministic: fixed seeds, CPU, torch 2.10)
Fixtures:
  office_a2d_graph.npz   shipped office A->D bridged graph (x, edge_index, y, masks)  [G5 data]
  conv_office.npz        G1: AdaptedConv(256,64) on the undirected office graph
  conv_small_D*.npz      G1: 200-node multigraph w/ isolated nodes, D in {2,31,64,128}
  ktgnn_office.npz       G2: KTGNN_no_complement(256,31,2,64,use_bn) eval forward, office graph
  ktgnn_sync.npz         G2: same on the 10k-node C2 synthetic graph (row subset + sums)
  partition_office.npz   G3: graph_partition / coalesce / to_undirected outputs
  knn_office_a2d.npz, knn_office_a2w.npz   G4(i)+G5: shipped ckpts (v2 / mlp scorer)
  knn_cosine_v1.npz      G4(ii): v1 cosine scorer (twitter ckpt) on seeded synthetic features
  knn_gauss.npz          G4(iii): raw Gaussian d=128 cosine (Similar_noTrans), Ns=20k, Nt=2k
  sage_encoder_v1.npz    8(f) rank 3: v1 GraphEncoder (2x SAGEConv, mean aggr) outputs on random graphs
  filters_office_a2d.npz 8(f) rank 2: check_added_edges_{cross,within}_domain_validity in/out (office A->D)
Round 2 (run alone with `--only c3,layers,a4,assembly`; the sections above are untouched and still reproduce bit for bit):
  ktgnn_c3.npz           BASELINE config 3: KTGNN_no_complement(300,2,2,128,use_bn) eval forward on synth.twitter_standin
  ktgnn_layers.npz       layer_num = 1 (KTGNN.py:344-347) and layer_num = 3 (:348-358) eval forwards on a small graph
  a4_office_a2d.npz      a4 glue: v2 encoders / class probs / get_probs_{cross,within}_domain on enumerated pairs, PairNorm modes
  assembly_office_a2d.npz  merge_graphs (main_bridged_graph.py:163-193) and reorder (:195-222) outputs on the office pieces
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import  # noqa: E402
from bridged_gnn_amd import synth  # noqa: E402
from bridged_gnn_amd.data import load_bridged_graph  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF = ref_import.REF_ROOT
torch.set_num_threads(8)


def sd_np(module, prefix=""):
    return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}: {os.path.getsize(path)/1e6:.2f} MB")


def randomize_bn(module, gen):
    """Non-trivial running stats / affine so eval-mode BN is actually exercised."""
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=gen) * 0.1)


def office_pieces(MD, SData, tag, norm_mode="None"):
    """(shipped graph, source Data, target Data, loaded Adversarial_Learner_v2, its state_dict, ns, nt) for an office pair"""
    name = {"a2d": "office_amazon2dslr", "a2w": "office_amazon2webcam"}[tag]
    dd = load_bridged_graph(os.path.join(REF, f"data_bridged_graph/{name}_bridged_graph.dat"))
    cmk = dd.central_mask
    ns = int(cmk.sum())
    nt = dd.x.shape[0] - ns
    loops = lambda n: torch.arange(n).unsqueeze(0).repeat(2, 1)
    ds = SData(x=dd.x[:ns], edge_index=loops(ns), y=dd.y[:ns].clone(), train_mask=dd.train_mask[:ns])
    dt = SData(x=dd.x[ns:], edge_index=loops(nt), y=dd.y[ns:].clone(),
               train_mask=dd.train_mask[ns:], val_mask=dd.val_mask[ns:], test_mask=dd.test_mask[ns:])
    ds_ctor = SData(x=ds.x, edge_index=ds.edge_index, y=torch.clamp(ds.y, min=0))
    ds_ctor.y[0] = 30
    dt_ctor = SData(x=dt.x, edge_index=dt.edge_index, y=torch.clamp(dt.y, min=0))
    dt_ctor.y[0] = 30
    sim = MD.Adversarial_Learner_v2(ds_ctor, dt_ctor, dim_hidden=128, num_layer=2, use_norm=True, source_clf=True,
                                    norm_mode=norm_mode, norm_scale=1.0, sim_mode="mlp", backbone="mlp")
    sd = torch.load(os.path.join(REF, f"ckpt/model_AdvLearner_{name}_best.ckpt"), map_location="cpu", weights_only=True)
    sim.load_state_dict(sd, strict=True)
    sim.eval()
    return dd, ds, dt, sim, sd, ns, nt


def round2_sections(KT, MD, BG, want):
    from torch_geometric.data import Data as SData
    from torch_geometric.utils import to_undirected as s_to_undirected

    if want("c3"):
        print("[C3] KTGNN on the Twitter stand-in (F=300, hidden=128)")
        xs, ei_s, ys, ms = synth.twitter_standin(seed=0)
        n = xs.shape[0]
        ei_u = s_to_undirected(torch.from_numpy(ei_s), num_nodes=n)
        torch.manual_seed(3)
        model = KT.KTGNN_no_complement(300, 2, 2, 128, root_weight=False, use_bn=True, dim_share=300, need_complement=False)
        randomize_bn(model, torch.Generator().manual_seed(9))
        model.eval()
        data = SData(x=torch.from_numpy(xs), edge_index=ei_u, y=torch.from_numpy(ys), central_mask=torch.from_numpy(ms))
        with torch.no_grad():
            lb, lt, lth, _ = model(data)
            emb = model.get_emb(data)
        rows = np.unique(np.concatenate([np.arange(0, 581, 4), np.arange(581, n, 16)]))
        save("ktgnn_c3.npz", rows=rows, logp_base=lb.numpy()[rows], logp_target=lt.numpy()[rows],
             logp_target_hat=lth.numpy()[rows], emb_rows=emb.numpy()[rows],
             sums=np.array([lb.double().sum().item(), lt.double().sum().item(), lth.double().sum().item()]),
             n_edges_undirected=np.int64(ei_u.shape[1]), edge_hash=np.int64(synth.edge_hash(ei_s, n)),
             **sd_np(model, "sd."))

    if want("layers"):
        print("[a15] layer_num in {1, 3}")
        xs, ei_s, ys, ms = synth.sync_rd_intra(n=1500, feat=32, homophily=0.7, deg=6, k_cross=8, seed=5)
        ei_u = s_to_undirected(torch.from_numpy(ei_s), num_nodes=1500)
        data = SData(x=torch.from_numpy(xs), edge_index=ei_u, y=torch.from_numpy(ys), central_mask=torch.from_numpy(ms))
        out = {}
        # layer_num = 1 (KTGNN.py:344-347): convs = [AdaptedConv(dim_in, num_classes)] feeding clf_* built for `hidden`
        # inputs -> only consistent when hidden == num_classes, and bns stays empty -> use_bn must be False
        torch.manual_seed(11)
        m1 = KT.KTGNN_no_complement(32, 4, 1, 4, root_weight=False, use_bn=False, dim_share=32, need_complement=False)
        randomize_bn(m1, torch.Generator().manual_seed(12))
        m1.eval()
        torch.manual_seed(13)
        m3 = KT.KTGNN_no_complement(32, 3, 3, 32, root_weight=False, use_bn=True, dim_share=32, need_complement=False)
        randomize_bn(m3, torch.Generator().manual_seed(14))
        m3.eval()
        with torch.no_grad():
            for tag, m in (("l1", m1), ("l3", m3)):
                lb, lt, lth, _ = m(data)
                out.update({f"{tag}.logp_base": lb.numpy(), f"{tag}.logp_target": lt.numpy(), f"{tag}.logp_target_hat": lth.numpy(),
                            f"{tag}.emb": m.get_emb(data).numpy()})
                out.update(sd_np(m, f"{tag}.sd."))
        save("ktgnn_layers.npz", **out)

    if want("a4"):
        print("[a4] v2 encoders / get_probs_* glue on office A->D")
        dd, ds, dt, sim, sd, ns, nt = office_pieces(MD, SData, "a2d")
        rng = np.random.default_rng(41)
        P = 600
        c_i1, c_i2 = torch.from_numpy(rng.integers(0, ns, P)), torch.from_numpy(rng.integers(0, nt, P))
        s_i1, s_i2 = torch.from_numpy(rng.integers(0, ns, P)), torch.from_numpy(rng.integers(0, ns, P))
        t_i1, t_i2 = torch.from_numpy(rng.integers(0, nt, P)), torch.from_numpy(rng.integers(0, nt, P))
        with torch.no_grad():
            pc, pcs, pct, zs, zt = sim.get_probs_cross_domain(ds, dt, c_i1, c_i2, return_representation=True)   # models.py:1132-1142
            pws, pws_clf = sim.get_probs_within_domain(ds, s_i1, s_i2, domain="source")                         # :1122-1131
            pwt, pwt_clf = sim.get_probs_within_domain(dt, t_i1, t_i2, domain="target")
        assert np.array_equal(zs.numpy(), np.load(os.path.join(OUT, "knn_office_a2d.npz"))["z_src"])
        arrs = dict(cross_idx1=c_i1.numpy(), cross_idx2=c_i2.numpy(), cross_probs=pc.numpy(), probs_clf_src=pcs.numpy(),
                    probs_clf_tar=pct.numpy(),
                    src_idx1=s_i1.numpy(), src_idx2=s_i2.numpy(), src_probs=pws.numpy(), src_probs_clf=pws_clf.numpy(),
                    tar_idx1=t_i1.numpy(), tar_idx2=t_i2.numpy(), tar_probs=pwt.numpy(), tar_probs_clf=pwt_clf.numpy())
        # PairNorm modes (models.py:29-64): the same weights under each mode (PairNorm has no parameters)
        for mode in ("PN", "PN-SI", "PN-SCS"):
            _, ds2, dt2, sim2, _, _, _ = office_pieces(MD, SData, "a2d", norm_mode=mode)
            with torch.no_grad():
                z2s = sim2.source_learner.backbone(ds2.x, ds2.edge_index)
                z2t, _ = sim2.target_learner.encode(dt2)
            arrs[f"z_src_{mode}"] = z2s.numpy()[::8]
            arrs[f"z_tar_{mode}"] = z2t.numpy()[::2]
        for k, v in sd.items():      # encoder weights (the scorer's sim_net.* and the reference's z are in knn_office_a2d.npz)
            if k.startswith(("source_learner.backbone.", "target_learner.equavilent_trans_layer.", "target_learner.encoder.")):
                arrs["sd." + k] = v.numpy()
        save("a4_office_a2d.npz", **arrs)

    if want("assembly"):
        print("[a9 / f4] merge_graphs + reorder on the office A->D pieces")
        import copy
        dd, ds, dt, sim, sd, ns, nt = office_pieces(MD, SData, "a2d")
        g = np.load(os.path.join(OUT, "knn_office_a2d.npz"))
        ec = torch.from_numpy(g["cross_edge_index"].astype(np.int64))
        es = torch.from_numpy(g["within_src_edge_index"].astype(np.int64))
        et = torch.from_numpy(g["within_tar_edge_index"].astype(np.int64))
        rng = np.random.default_rng(43)
        # "original" edges of the two domains: random, with duplicates and self loops (the office loaders use self loops only)
        ds.edge_index = torch.from_numpy(np.concatenate([rng.integers(0, ns, (2, 3000)), np.tile(np.arange(ns), (2, 1))], axis=1))
        dt.edge_index = torch.from_numpy(np.concatenate([rng.integers(0, nt, (2, 700)), np.tile(np.arange(nt), (2, 1))], axis=1))
        merged = BG.merge_graphs(ds, dt, copy.deepcopy(ec), es, et)                                   # :163-193 (mutates arg :170)
        arrs = dict(ei_src=ds.edge_index.numpy().astype(np.int32), ei_tar=dt.edge_index.numpy().astype(np.int32),
                    merged_edge_index=merged.edge_index.numpy().astype(np.int32), merged_y=merged.y.numpy().astype(np.int16),
                    merged_train=merged.train_mask.numpy(), merged_val=merged.val_mask.numpy(),
                    merged_test=merged.test_mask.numpy(), merged_central=merged.central_mask.numpy(),
                    merged_x_col0=merged.x[:, 0].numpy())
        perm = rng.permutation(ns + nt)
        orig_src, orig_tar = perm[:ns], perm[ns:]
        m_src = {int(o): i for i, o in enumerate(orig_src)}          # orig id -> local index (utils.py:58-63)
        m_tar = {int(o): i for i, o in enumerate(orig_tar)}
        ro = BG.reorder(copy.deepcopy(merged), ds, m_src, m_tar)                                      # :195-222
        arrs.update(orig_src=orig_src.astype(np.int32), orig_tar=orig_tar.astype(np.int32),
                    reordered_edge_index=ro.edge_index.numpy().astype(np.int32), reordered_y=ro.y.numpy().astype(np.int16),
                    reordered_train=ro.train_mask.numpy(), reordered_val=ro.val_mask.numpy(), reordered_test=ro.test_mask.numpy(),
                    reordered_central=ro.central_mask.numpy(), reordered_x_col0=ro.x[:, 0].numpy())
        save("assembly_office_a2d.npz", **arrs)


def round3_sections(want):
    """Round 3 (`--only f4`): SURVEY 8(f) rank 4 leftovers -- the reference's `utils.dataset_conversion` (:41-99, both split modes),
    `eval_bridged_Graph` (:101-113) and `eval_homophily` (:115-131; it only PRINTS its two ratios: they are parsed from stdout) run
    on a seeded synthetic VS-graph under the shim (oracle/shim/torch_sparse: SparseTensor / matmul restated)."""
    if not want("f4"):
        return
    import contextlib
    import io
    import utils as RU                       # the reference's utils.py (sys.path set by ref_import)
    from torch_geometric.data import Data as SData
    print("[f4] dataset_conversion / eval_bridged_Graph / eval_homophily")
    rng = np.random.default_rng(2024)
    n, F_, C = 700, 12, 3
    cm = rng.random(n) < 0.45
    x = rng.standard_normal((n, F_)).astype(np.float32)
    y = rng.integers(-1, C, n).astype(np.int64)                    # -1 = unlabeled
    ei = rng.integers(0, n, (2, 4200)).astype(np.int64)            # within- and cross-domain edges, duplicates, self loops
    tr = rng.random(n) < 0.5
    va = ~tr & (rng.random(n) < 0.5)
    te = ~tr & ~va

    def mk():
        return SData(x=torch.from_numpy(x), edge_index=torch.from_numpy(ei), y=torch.from_numpy(y),
                     central_mask=torch.from_numpy(cm), train_mask=torch.from_numpy(tr), val_mask=torch.from_numpy(va),
                     test_mask=torch.from_numpy(te))
    arrs = dict(x=x, y=y.astype(np.int16), edge_index=ei.astype(np.int32), central_mask=cm, train_mask=tr, val_mask=va, test_mask=te)
    for tag, split in (("split", True), ("keep", False)):
        with contextlib.redirect_stdout(io.StringIO()):
            ds, dt, ms, mt = RU.dataset_conversion(mk(), seed=3, train_val_test_ratio=[0.6, 0.2, 0.2], dataset_name=None, split_data=split)
        inv_s = np.empty(len(ms), dtype=np.int32); inv_t = np.empty(len(mt), dtype=np.int32)
        for o, l in ms.items(): inv_s[l] = o
        for o, l in mt.items(): inv_t[l] = o
        for nm, d in (("src", ds), ("tar", dt)):
            arrs.update({f"{tag}_{nm}_x": d.x.numpy(), f"{tag}_{nm}_edge_index": d.edge_index.numpy().astype(np.int32),
                         f"{tag}_{nm}_y": d.y.numpy().astype(np.int16), f"{tag}_{nm}_train": d.train_mask.numpy(),
                         f"{tag}_{nm}_val": d.val_mask.numpy(), f"{tag}_{nm}_test": d.test_mask.numpy()})
        arrs.update({f"{tag}_orig_of_src": inv_s, f"{tag}_orig_of_tar": inv_t})
    # the 'twitter' branch (:45-49): target features cut to the first 300 columns
    xw = rng.standard_normal((60, 310)).astype(np.float32)
    cw = rng.random(60) < 0.5
    dw = SData(x=torch.from_numpy(xw), edge_index=torch.from_numpy(rng.integers(0, 60, (2, 200)).astype(np.int64)),
               y=torch.from_numpy(rng.integers(0, 2, 60).astype(np.int64)), central_mask=torch.from_numpy(cw),
               train_mask=torch.zeros(60, dtype=torch.bool), val_mask=torch.zeros(60, dtype=torch.bool), test_mask=torch.zeros(60, dtype=torch.bool))
    with contextlib.redirect_stdout(io.StringIO()):
        ds, dt, _, _ = RU.dataset_conversion(dw, seed=1, dataset_name="twitter")
    arrs.update(tw_x=xw, tw_central=cw, tw_src_shape=np.array(ds.x.shape), tw_tar_shape=np.array(dt.x.shape))
    # evaluation helpers on the whole synthetic graph
    g = mk()
    with contextlib.redirect_stdout(io.StringIO()):
        ratio = RU.eval_bridged_Graph(g)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        RU.eval_homophily(g)
    vals = [float(l.split(":")[1]) for l in buf.getvalue().splitlines() if ":" in l]
    arrs.update(eval_bridged_ratio=np.float64(float(ratio)), homophily_1st=np.float64(vals[0]), homophily_2nd=np.float64(vals[1]))
    print(f"  eval_bridged_Graph {float(ratio):.6f}, homophily {vals[0]:.6f} / 2nd order {vals[1]:.6f}")
    save("f4_utils.npz", **arrs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="comma list of round-2 / round-3 sections (c3,layers,a4,assembly,f4); default: everything")
    only = [t for t in ap.parse_args().only.split(",") if t]
    os.makedirs(OUT, exist_ok=True)
    KT, MD, BG = ref_import.import_reference()
    if only:
        round2_sections(KT, MD, BG, lambda t: t in only)
        round3_sections(lambda t: t in only)
        print("done")
        return
    import torch_geometric
    from torch_geometric.data import Data as SData
    from torch_geometric.utils import coalesce as s_coalesce, to_undirected as s_to_undirected

    # ------------------------------------------------------------------ shipped office A->D graph
    print("[G5] shipped office graphs")
    d = load_bridged_graph(os.path.join(REF, "data_bridged_graph/office_amazon2dslr_bridged_graph.dat"))
    x, ei, cm = d.x, d.edge_index, d.central_mask
    assert bool(cm[:2817].all()) and not bool(cm[2817:].any())
    save("office_a2d_graph.npz", x=x.numpy(), edge_index=ei.numpy().astype(np.int32), y=d.y.numpy(),
         train_mask=d.train_mask.numpy(), val_mask=d.val_mask.numpy(), test_mask=d.test_mask.numpy(),
         central_mask=cm.numpy())
    ei_und = s_to_undirected(ei, num_nodes=x.shape[0])

    # ------------------------------------------------------------------ G3 partition
    print("[G3] graph_partition / coalesce / to_undirected")
    model = KT.KTGNN_no_complement(256, 31, 2, 64, root_weight=False, use_bn=True, dim_share=256,
                                   need_complement=False)
    e1, e2, ecat = model.graph_partition(ei_und, cm)
    e1d, e2d, _ = model.graph_partition(ei, cm)
    save("partition_office.npz", ei_undirected=ei_und.numpy().astype(np.int32),
         e1=e1.numpy().astype(np.int32), e2=e2.numpy().astype(np.int32),
         e1_directed=e1d.numpy().astype(np.int32), e2_directed=e2d.numpy().astype(np.int32))

    # ------------------------------------------------------------------ G1 AdaptedConv on office
    print("[G1] AdaptedConv office")
    torch.manual_seed(0)
    conv = KT.AdaptedConv(256, 64, root_weight=False).eval()
    with torch.no_grad():
        # intermediate tensors by re-tracing the reference's own ops (KTGNN.py:275-299)
        out = conv(x, ecat, e1, e2, cm)
        xs = x
        dd = xs[cm].mean(0, keepdim=True) - xs[~cm].mean(0, keepdim=True)
        dd = dd.expand(xs.shape)
        s2t = torch.tanh(conv.a_g_s2t(torch.cat((xs, dd), -1))) * dd
        t2s = torch.tanh(conv.a_g_t2s(torch.cat((xs, dd), -1))) * dd
        h_s2t = conv.lin_t(xs - s2t * cm.unsqueeze(-1))
        h_t2s = conv.lin_s(xs + t2s * (~cm).unsqueeze(-1))
        a1 = conv.a_f_t2s(torch.nn.functional.leaky_relu(h_t2s[e1[0]] + h_t2s[e1[1]], 0.1))
        a2 = conv.a_f_s2t(torch.nn.functional.leaky_relu(h_s2t[e2[0]] + h_s2t[e2[1]], 0.1))
        alpha = torch_geometric.utils.softmax(torch.cat((a1, a2), 0), ecat[1], num_nodes=x.shape[0])
    save("conv_office.npz", out=out.numpy(), alpha=alpha.numpy().reshape(-1),
         h_s2t_rows=h_s2t[::8].numpy(), h_t2s_rows=h_t2s[::8].numpy(), **sd_np(conv, "p."))

    # ------------------------------------------------------------------ G1 small multigraphs
    print("[G1] AdaptedConv small multigraphs")
    for D in (2, 31, 64, 128):
        gen = torch.Generator().manual_seed(100 + D)
        n, din = 200, 48
        ei_s, m_s = synth.random_multigraph(n, 1500, frac_src=0.45, n_isolated=7, seed=D)
        ei_t, m_t = torch.from_numpy(ei_s), torch.from_numpy(m_s)
        xs = torch.randn(n, din, generator=gen)
        torch.manual_seed(D)
        conv = KT.AdaptedConv(din, D, root_weight=False).eval()
        mm = KT.KTGNN_no_complement(din, 2, 2, 8, dim_share=din)
        f1, f2, fc = mm.graph_partition(ei_t, m_t)
        with torch.no_grad():
            o = conv(xs, fc, f1, f2, m_t)
        save(f"conv_small_D{D}.npz", x=xs.numpy(), edge_index=ei_s.astype(np.int32), central_mask=m_s,
             out=o.numpy(), **sd_np(conv, "p."))

    # ------------------------------------------------------------------ G2 KTGNN office
    print("[G2] KTGNN office")
    torch.manual_seed(0)
    model = KT.KTGNN_no_complement(256, 31, 2, 64, root_weight=False, use_bn=True, dim_share=256,
                                   need_complement=False)
    randomize_bn(model, torch.Generator().manual_seed(7))
    model.eval()
    data = SData(x=x, edge_index=ei_und, y=d.y, central_mask=cm)
    with torch.no_grad():
        lb, lt, lth, _ = model(data)
        emb = model.get_emb(data)
    save("ktgnn_office.npz", logp_base=lb.numpy(), logp_target=lt.numpy(), logp_target_hat=lth.numpy(),
         emb_rows=emb[::8].numpy(), **sd_np(model, "sd."))

    # ------------------------------------------------------------------ G2 KTGNN on C2 synthetic
    print("[G2] KTGNN sync-rd (C2)")
    xs, ei_s, ys, ms = synth.sync_rd_intra(n=10000, feat=64, homophily=0.7, deg=10, k_cross=20, seed=0)
    ei_u = s_to_undirected(torch.from_numpy(ei_s), num_nodes=10000)
    torch.manual_seed(1)
    model = KT.KTGNN_no_complement(64, 2, 2, 64, root_weight=False, use_bn=True, dim_share=64,
                                   need_complement=False)
    randomize_bn(model, torch.Generator().manual_seed(8))
    model.eval()
    data = SData(x=torch.from_numpy(xs), edge_index=ei_u, y=torch.from_numpy(ys),
                 central_mask=torch.from_numpy(ms))
    with torch.no_grad():
        lb, lt, lth, _ = model(data)
    rows = np.arange(0, 10000, 16)
    save("ktgnn_sync.npz", rows=rows, logp_base=lb.numpy()[rows], logp_target=lt.numpy()[rows],
         logp_target_hat=lth.numpy()[rows],
         sums=np.array([lb.double().sum().item(), lt.double().sum().item(), lth.double().sum().item()]),
         n_edges_undirected=np.int64(ei_u.shape[1]), **sd_np(model, "sd."))

    # ------------------------------------------------------------------ G4(i)/G5 office kNN
    for tag, k_cross in (("a2d", 20), ("a2w", 8)):
        name = {"a2d": "office_amazon2dslr", "a2w": "office_amazon2webcam"}[tag]
        print(f"[G4/G5] kNN {name}")
        dd = load_bridged_graph(os.path.join(REF, f"data_bridged_graph/{name}_bridged_graph.dat"))
        cmk = dd.central_mask
        ns = int(cmk.sum())
        assert bool(cmk[:ns].all())
        nt = dd.x.shape[0] - ns
        loops = lambda n: torch.arange(n).unsqueeze(0).repeat(2, 1)
        ds = SData(x=dd.x[:ns], edge_index=loops(ns), y=dd.y[:ns].clone())
        dt = SData(x=dd.x[ns:], edge_index=loops(nt), y=dd.y[ns:].clone(),
                   train_mask=dd.train_mask[ns:], val_mask=dd.val_mask[ns:], test_mask=dd.test_mask[ns:])
        # num_classes is derived from y.max() in the reference ctor (models.py:1004); the shipped
        # target y contains -1 for unlabeled nodes, classes = 31
        ds_ctor = SData(x=ds.x, edge_index=ds.edge_index, y=torch.clamp(ds.y, min=0))
        ds_ctor.y[0] = 30
        dt_ctor = SData(x=dt.x, edge_index=dt.edge_index, y=torch.clamp(dt.y, min=0))
        dt_ctor.y[0] = 30
        sim = MD.Adversarial_Learner_v2(ds_ctor, dt_ctor, dim_hidden=128, num_layer=2, use_norm=True,
                                        source_clf=True, norm_mode="None", norm_scale=1.0,
                                        sim_mode="mlp", backbone="mlp")
        sd = torch.load(os.path.join(REF, f"ckpt/model_AdvLearner_{name}_best.ckpt"),
                        map_location="cpu", weights_only=True)
        sim.load_state_dict(sd, strict=True)
        sim.eval()
        with torch.no_grad():
            ec, esim, eidx, pcs, pct = BG.add_topk_sim_cross_domain_edges(ds, dt, sim, k=k_cross, batch_size=100)
            z_src = sim.source_learner.backbone(ds.x, ds.edge_index)
            z_tar, _ = sim.target_learner.encode(dt)
            es, esims, eidxs = BG.add_topk_sim_within_domain_edges(ds, sim, k=3, batch_size=100, domain="source")
            et, esimt, eidxt = BG.add_topk_sim_within_domain_edges(dt, sim, k=3, batch_size=100, domain="target")
        if tag == "a2d":
            # [8(f) rank 2] the reference's own validity filters on its own (misaligned) inputs
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
                fc = BG.check_added_edges_cross_domain_validity(ec, esim.view(-1), ds, dt, pcs, pct,
                                                                thres_conf_quantile=0.1, thres_feat_sim=0.8)
                ds_w = SData(x=ds.x, y=ds.y, train_mask=dd.train_mask[:ns])
                fw = BG.check_added_edges_within_domain_validity(es, esims.view(-1), ds_w, pcs,
                                                                 thres_conf_quantile=0.1, thres_feat_sim=0.8)
            save("filters_office_a2d.npz", cross_in=ec.numpy().astype(np.int32), cross_e_sim_flat=esim.view(-1).numpy(),
                 cross_out=fc.numpy().astype(np.int32), within_in=es.numpy().astype(np.int32),
                 within_e_sim_flat=esims.view(-1).numpy(), within_out=fw.numpy().astype(np.int32),
                 probs_clf_src=pcs.numpy(), probs_clf_tar=pct.numpy(), train_mask_src=dd.train_mask[:ns].numpy(),
                 train_mask_tar=dd.train_mask[ns:].numpy())
        simnet = {"sim." + k[len("source_learner.sim_net."):]: v.numpy() for k, v in sd.items()
                  if k.startswith("source_learner.sim_net.")}
        save(f"knn_office_{tag}.npz", z_src=z_src.numpy(), z_tar=z_tar.numpy(), k_cross=np.int64(k_cross),
             cross_edge_index=ec.numpy().astype(np.int32), cross_e_sim=esim.numpy(),
             cross_idx=eidx.numpy().astype(np.int32),
             pred_clf_src=pcs.argmax(1).numpy().astype(np.int16), pred_clf_tar=pct.argmax(1).numpy().astype(np.int16),
             probs_clf_src_rows=pcs[::16].numpy(), probs_clf_tar_rows=pct[::16].numpy(),
             within_src_edge_index=es.numpy().astype(np.int32), within_src_e_sim=esims.numpy(),
             within_src_idx=eidxs.numpy().astype(np.int32),
             within_tar_edge_index=et.numpy().astype(np.int32), within_tar_e_sim=esimt.numpy(),
             within_tar_idx=eidxt.numpy().astype(np.int32),
             shipped_edge_index=dd.edge_index.numpy().astype(np.int32), n_src=np.int64(ns),
             y=dd.y.numpy().astype(np.int16), **simnet)

    # ------------------------------------------------------------------ G4(ii) v1 cosine scorer
    print("[G4] v1 cosine scorer (twitter ckpt, synthetic features)")
    gen = torch.Generator().manual_seed(11)
    ns = nt = 2000
    F_in = 300
    loops = lambda n: torch.arange(n).unsqueeze(0).repeat(2, 1)
    ds = SData(x=torch.randn(ns, F_in, generator=gen) * 0.5, edge_index=loops(ns),
               y=torch.randint(0, 2, (ns,), generator=gen))
    dt = SData(x=torch.randn(nt, F_in, generator=gen) * 0.5 + 0.1, edge_index=loops(nt),
               y=torch.randint(0, 2, (nt,), generator=gen))
    sim = MD.Adversarial_Learner(ds, dt, dim_hidden=64, num_layer=2, source_clf=True, norm_mode="None", norm_scale=1.0)
    sd = torch.load(os.path.join(REF, "ckpt/model_AdvLearner_twitter_unrelational_best.ckpt"),
                    map_location="cpu", weights_only=True)
    sim.load_state_dict(sd, strict=True)
    sim.eval()
    with torch.no_grad():
        ec, esim, eidx, pcs, pct = BG.add_topk_sim_cross_domain_edges(ds, dt, sim, k=20, batch_size=100)
        z_src = sim.source_learner.backbone(ds.x, ds.edge_index)
        z_tar, _ = sim.target_learner.encode(dt)
        sn = sim.source_learner.sim_net
        u_s, u_t = sn.lin_self(z_src), sn.lin_self(z_tar)
        q_src, q_tar = u_s + sn.biasatt(u_s), u_t + sn.biasatt(u_t)
    simnet = {"sim." + k[len("source_learner.sim_net."):]: v.numpy() for k, v in sd.items()
              if k.startswith("source_learner.sim_net.")}
    save("knn_cosine_v1.npz", z_src=z_src.numpy(), z_tar=z_tar.numpy(), q_src=q_src.numpy(), q_tar=q_tar.numpy(),
         cross_edge_index=ec.numpy().astype(np.int32), cross_e_sim=esim.numpy(),
         cross_idx=eidx.numpy().astype(np.int32), **simnet)

    # ------------------------------------------------------------------ 8(f) rank 3: v1 SAGEConv encoders
    print("[8f-3] v1 SAGEConv encoders on random graphs (twitter ckpt)")
    gen = torch.Generator().manual_seed(31)
    n_s, n_t = 320, 280
    ei_s = torch.randint(0, n_s, (2, 1500), generator=gen)
    ei_t = torch.randint(0, n_t, (2, 900), generator=gen)
    ei_t = ei_t[:, ei_t[1] < n_t - 5]                       # a few target nodes without in-edges
    xs = torch.randn(n_s, F_in, generator=gen) * 0.5
    xt = torch.randn(n_t, F_in, generator=gen) * 0.5 + 0.1
    with torch.no_grad():
        zs = sim.source_learner.backbone(xs, ei_s)
        zt, _ = sim.target_learner.encode(SData(x=xt, edge_index=ei_t))
    enc = {"sd." + k: v.numpy() for k, v in sd.items() if k.startswith(("source_learner.backbone.", "target_learner.encoder.",
                                                                        "target_learner.equavilent_trans_layer."))}
    save("sage_encoder_v1.npz", x_src=xs.numpy(), x_tar=xt.numpy(), ei_src=ei_s.numpy().astype(np.int32),
         ei_tar=ei_t.numpy().astype(np.int32), z_src=zs.numpy(), z_tar=zt.numpy(), **enc)

    # ------------------------------------------------------------------ G4(iii) raw Gaussian cosine
    print("[G4] raw Gaussian cosine via Similar_noTrans")
    ns, nt, dimq, k = 20000, 2000, 128, 20
    q_src = torch.from_numpy(synth.gaussian_embeddings(ns, dimq, seed=21))
    q_tar = torch.from_numpy(synth.gaussian_embeddings(nt, dimq, seed=22))
    snt = MD.Similar_noTrans(dimq, 2, use_clf=False).eval()
    all_src = torch.arange(ns).unsqueeze(-1)
    vals, idxs = [], []
    with torch.no_grad():
        for s in range(0, nt, 50):
            bt = torch.arange(s, min(s + 50, nt)).unsqueeze(-1)
            pairs = MD.pair_enumeration(all_src, bt).transpose(0, 1)          # main_bridged_graph.py:49
            p = snt.similarity_cross_domain(q_src, q_tar, pairs[0], pairs[1])  # models.py:185-189
            tk = p.view(-1, ns).topk(k=k, dim=1, largest=True, sorted=False)   # main_bridged_graph.py:59-60
            vals.append(tk.values)
            idxs.append(tk.indices)
    vals, idxs = torch.cat(vals), torch.cat(idxs)
    save("knn_gauss.npz", ns=np.int64(ns), nt=np.int64(nt), d=np.int64(dimq), k=np.int64(k),
         seed_src=np.int64(21), seed_tar=np.int64(22), e_sim=vals.numpy(), idx=idxs.numpy().astype(np.int32))
    round2_sections(KT, MD, BG, lambda t: True)
    round3_sections(lambda t: True)
    print("done")


if __name__ == "__main__":
    main()
