"""CPU: the plain-C oracle (oracle/oracle_c.c) against the numpy oracle (pinned to the reference's
golden vectors by test_oracle_golden.py).  Canonical (fp64, index-order) routines must agree
BIT-FOR-BIT between C and numpy; fp32 routines within the 1e-5 parity bar."""
import numpy as np

from conftest import assert_close, sub
from oracle import oracle_c as OC
from oracle import oracle_np as O


def test_c_transform_and_aggregate_office(golden):
    g, p, c = golden("office_a2d_graph.npz"), golden("partition_office.npz"), golden("conv_office.npz")
    prm = sub(c, "p.")
    hs2t, ht2s = OC.adaptedconv_transform(g["x"], g["central_mask"], prm)
    assert_close(hs2t[::8], c["h_s2t_rows"], what="h_s2t")
    assert_close(ht2s[::8], c["h_t2s_rows"], what="h_t2s")
    rowptr, col, eperm = O.dst_csr(p["ei_undirected"], g["central_mask"])
    out, alpha = OC.adaptedconv_aggregate(ht2s, hs2t, prm["a_f_t2s.weight"], prm["a_f_s2t.weight"],
                                          rowptr, col, g["central_mask"], want_alpha=True)
    assert_close(out, c["out"], what="out")
    # alpha: map CSR order back to the reference's cat(E1,E2) order
    m = g["central_mask"]
    rew = np.concatenate([p["ei_undirected"][:, p["ei_undirected"][0] != p["ei_undirected"][1]],
                          np.stack([np.arange(len(m)), np.arange(len(m))])], axis=1)
    pos_in_cat = np.empty(rew.shape[1], np.int64)
    d_in_s = m[rew[1]]
    pos_in_cat[np.nonzero(d_in_s)[0]] = np.arange(d_in_s.sum())
    pos_in_cat[np.nonzero(~d_in_s)[0]] = d_in_s.sum() + np.arange((~d_in_s).sum())
    assert_close(alpha, c["alpha"][pos_in_cat[eperm]], what="alpha")


def test_c_canonical_bit_exact_vs_numpy():
    rng = np.random.default_rng(3)
    q = rng.standard_normal((37, 128)).astype(np.float32)
    c = rng.standard_normal((501, 128)).astype(np.float32)
    qn, cn = O.l2_normalize_rows(q), O.l2_normalize_rows(c)
    assert np.array_equal(qn, OC.l2_normalize_rows(q)) and np.array_equal(cn, OC.l2_normalize_rows(c))
    v_np, i_np = O.topk_rows(O.cosine_scores_canonical(qn, cn), 20)
    v_c, i_c = OC.cosine_topk(qn, cn, 20)
    assert np.array_equal(i_np, i_c) and np.array_equal(v_np, v_c)
    # exact ties (duplicated candidates): lower index wins in both
    c2 = np.concatenate([cn, cn[:50]])
    v_np, i_np = O.topk_rows(O.cosine_scores_canonical(qn, c2), 20)
    v_c, i_c = OC.cosine_topk(qn, c2, 20)
    assert np.array_equal(i_np, i_c) and np.array_equal(v_np, v_c)


def test_c_mlp_topk_bit_exact_vs_numpy(golden):
    f = golden("knn_office_a2d.npz")
    A, B, scale, shift, w2, b2 = O.mlp_pair_terms(f["z_src"], f["z_tar"][:64], sub(f, "sim."))
    v_np, i_np = O.topk_rows(O.mlp_scores_canonical(A, B, scale, shift, w2, b2), 20)
    v_c, i_c = OC.mlp_topk(A, B, scale, shift, w2, b2, 20)
    assert np.array_equal(i_np, i_c) and np.array_equal(v_np, v_c)


def test_c_ktgnn_forward_matches_numpy_and_golden(golden):
    """the C composition of the whole eval forward (used for the full-size parity samples and bench.py's checksums)
    against the numpy restatement and the reference's office golden"""
    g, p, k = golden("office_a2d_graph.npz"), golden("partition_office.npz"), golden("ktgnn_office.npz")
    sd = sub(k, "sd.")
    rowptr, col, _ = O.dst_csr(p["ei_undirected"], g["central_mask"])
    lb, lt, lth, emb = OC.ktgnn_forward_eval(g["x"], rowptr, col, g["central_mask"], sd, return_emb=True)
    nb, nt, nth = O.ktgnn_forward_eval(g["x"], p["ei_undirected"], g["central_mask"], sd)
    for a, b, c, what in ((lb, nb, k["logp_base"], "base"), (lt, nt, k["logp_target"], "target"), (lth, nth, k["logp_target_hat"], "target_hat")):
        assert_close(a, b, what=f"C vs numpy {what}")
        assert_close(a, c, what=f"C vs reference {what}")
    assert_close(emb[::8], k["emb_rows"], what="emb")
