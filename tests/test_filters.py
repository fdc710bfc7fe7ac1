"""Edge validity filters (reference main_bridged_graph.py:123-161, :225-264): bit-exact
vs the reference's own output on the shipped office A->D artefacts, incl. its index-misalignment quirk."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def _setup(dev):
    from bridged_gnn_amd.data import Data
    f, g, k = load_golden("filters_office_a2d.npz"), load_golden("office_a2d_graph.npz"), load_golden("knn_office_a2d.npz")
    ns = 2817
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    y = t(g["y"])
    ds = Data(x=t(g["x"][:ns]), y=y[:ns], train_mask=t(f["train_mask_src"]))
    dt = Data(x=t(g["x"][ns:]), y=y[ns:], train_mask=t(f["train_mask_tar"]))
    return f, k, ds, dt, t


def _run(dev):
    from bridged_gnn_amd import bridge
    f, k, ds, dt, t = _setup(dev)
    pcs, pct = t(f["probs_clf_src"]), t(f["probs_clf_tar"])
    # (1) reference behaviour reproduced bit for bit when fed the reference's own (misaligned) e_sim vector
    out = bridge.check_added_edges_cross_domain_validity(t(f["cross_in"].astype(np.int64)), t(f["cross_e_sim_flat"]), ds, dt,
                                                         pcs, pct, thres_conf_quantile=0.1, thres_feat_sim=0.8)
    assert np.array_equal(out.cpu().numpy(), f["cross_out"])
    out_w = bridge.check_added_edges_within_domain_validity(t(f["within_in"].astype(np.int64)), t(f["within_e_sim_flat"]), ds,
                                                            pcs, thres_conf_quantile=0.1, thres_feat_sim=0.8)
    assert np.array_equal(out_w.cpu().numpy(), f["within_out"])
    # (2) aligned semantics: every edge is paired with ITS OWN similarity
    ei = t(f["cross_in"].astype(np.int64))
    e_al = bridge.align_e_sim_to_edges(ei, t(k["cross_e_sim"]), t(k["cross_idx"].astype(np.int64)))
    idx = k["cross_idx"]
    for e in (0, 17, 4242, ei.shape[1] - 1):
        s, q = int(ei[0, e]), int(ei[1, e])
        assert e_al[e].item() == k["cross_e_sim"][q, list(idx[q]).index(s)]
    out_al = bridge.check_added_edges_cross_domain_validity(ei, e_al, ds, dt, pcs, pct, 0.1, 0.8)
    thr = np.quantile(e_al.cpu().numpy().astype(np.float64), 0.1)
    key_in = ei[0].cpu().numpy() * 10000 + ei[1].cpu().numpy()
    key_out = out_al[0].cpu().numpy() * 10000 + out_al[1].cpu().numpy()
    kept = e_al.cpu().numpy()[np.isin(key_in, key_out)]
    assert kept.shape[0] == out_al.shape[1] and (kept >= thr - 1e-6).all()   # no low-confidence edge survives


def test_filters_cpu_tensors():
    _run("cpu")


@pytest.mark.gpu
def test_filters_gpu_tensors():
    _run("cuda:0")
