"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/_build/liboracle.so
(plain-C restatement, see oracle_c.c).  Importers: tests/, __graft_entry__.smoke(), bench.py
(cpu_baseline leg)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "oracle_c.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def num_threads():
    return int(lib().orc_num_threads())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def adaptedconv_transform(x, mask, p):
    x = _f32(x)
    N, Din = x.shape
    Ws, Wt = _f32(p["lin_s.weight"]), _f32(p["lin_t.weight"])
    bs = _f32(p["lin_s.bias"]) if p.get("lin_s.bias") is not None else None
    bt = _f32(p["lin_t.bias"]) if p.get("lin_t.bias") is not None else None
    D = Ws.shape[0]
    g1, g2 = _f32(p["a_g_s2t.weight"]).reshape(-1), _f32(p["a_g_t2s.weight"]).reshape(-1)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    hs2t = np.empty((N, D), np.float32)
    ht2s = np.empty((N, D), np.float32)
    lib().orc_adaptedconv_transform_f32(_p(x), C.c_int64(N), C.c_int32(Din), _p(m), _p(Ws), _p(bs), _p(Wt),
                                        _p(bt), _p(g1), _p(g2), C.c_int32(D), _p(hs2t), _p(ht2s))
    return hs2t, ht2s


def adaptedconv_aggregate(h_t2s, h_s2t, a_t2s, a_s2t, rowptr, col, mask, slope=0.1, want_alpha=False):
    h_t2s, h_s2t = _f32(h_t2s), _f32(h_s2t)
    N, D = h_t2s.shape
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    cl = np.ascontiguousarray(col, dtype=np.int32)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.empty((N, D), np.float32)
    alpha = np.empty(cl.shape[0], np.float32) if want_alpha else None
    lib().orc_adaptedconv_aggregate_f32(_p(h_t2s), _p(h_s2t), _p(_f32(a_t2s).reshape(-1)), _p(_f32(a_s2t).reshape(-1)),
                                        _p(rp), _p(cl), _p(m), C.c_int64(N), C.c_int32(D), C.c_int64(D),
                                        C.c_float(slope), _p(out), C.c_int64(D), _p(alpha))
    return (out, alpha) if want_alpha else out


def adaptedconv_aggregate_f64(h_t2s, h_s2t, a_t2s, a_s2t, rowptr, col, mask, slope=0.1):
    """every intermediate in fp64 (inputs fp32): the real-arithmetic value of the reference's formula"""
    h_t2s, h_s2t = _f32(h_t2s), _f32(h_s2t)
    N, D = h_t2s.shape
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    cl = np.ascontiguousarray(col, dtype=np.int32)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.empty((N, D), np.float64)
    lib().orc_adaptedconv_aggregate_f64(_p(h_t2s), _p(h_s2t), _p(_f32(a_t2s).reshape(-1)), _p(_f32(a_s2t).reshape(-1)),
                                        _p(rp), _p(cl), _p(m), C.c_int64(N), C.c_int32(D), C.c_int64(D),
                                        C.c_float(slope), _p(out), C.c_int64(D))
    return out


def l2_normalize_rows(q, eps=1e-8):
    q = _f32(q)
    out = np.empty_like(q)
    lib().orc_l2_normalize_rows_f32(_p(q), C.c_int64(q.shape[0]), C.c_int32(q.shape[1]), C.c_float(eps), _p(out))
    return out


def cosine_topk(qn_query, qn_cand, k):
    a, b = _f32(qn_query), _f32(qn_cand)
    idx = np.empty((a.shape[0], k), np.int64)
    val = np.empty((a.shape[0], k), np.float64)
    lib().orc_cosine_topk(_p(a), _p(b), C.c_int64(a.shape[0]), C.c_int64(b.shape[0]), C.c_int32(a.shape[1]),
                          C.c_int32(k), _p(idx), _p(val))
    return val, idx


def cosine_topk_refshape(qn_query, qn_cand, k):
    a, b = _f32(qn_query), _f32(qn_cand)
    idx = np.empty((a.shape[0], k), np.int64)
    val = np.empty((a.shape[0], k), np.float32)
    lib().orc_cosine_topk_f32_refshape(_p(a), _p(b), C.c_int64(a.shape[0]), C.c_int64(b.shape[0]),
                                       C.c_int32(a.shape[1]), C.c_int32(k), _p(idx), _p(val))
    return val, idx


def mlp_topk(A_cand, B_query, scale, shift, w2, b2, k):
    A, B = _f32(A_cand), _f32(B_query)
    idx = np.empty((B.shape[0], k), np.int64)
    val = np.empty((B.shape[0], k), np.float64)
    lib().orc_mlp_topk(_p(A), _p(B), _p(_f32(scale)), _p(_f32(shift)), _p(_f32(w2)), C.c_float(float(b2)),
                       C.c_int64(B.shape[0]), C.c_int64(A.shape[0]), C.c_int32(A.shape[1]), C.c_int32(k),
                       _p(idx), _p(val))
    return val, idx


def ktgnn_forward_eval(x, rowptr, col, mask, sd, use_bn=True, return_emb=False):
    """KTGNN_no_complement.forward in eval mode (models/KTGNN.py:401-435, need_complement=False) composed from the C
    routines above over a by-destination CSR (oracle_np.dst_csr); same result as oracle_np.ktgnn_forward_eval, which
    is the one pinned to the reference's golden vectors (tests/test_oracle_c.py checks the two against each other).
    -> (logp_base, logp_target, logp_target_hat[, emb])."""
    from . import oracle_np as O

    def conv(xx, prefix):
        p = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        hs2t, ht2s = adaptedconv_transform(xx, mask, p)
        return adaptedconv_aggregate(ht2s, hs2t, p["a_f_t2s.weight"], p["a_f_s2t.weight"], rowptr, col, mask)

    def bn(h, prefix, eps=1e-5):
        b = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        return ((h - b["running_mean"]) / np.sqrt(b["running_var"] + np.float32(eps)) * b["weight"] + b["bias"]).astype(np.float32)

    h = _f32(x)
    i = 0
    while f"convs.{i}.lin_s.weight" in sd:                                     # :418-430
        h = conv(h, f"convs.{i}.")
        if use_bn:
            h = bn(h, f"bns.{i}.")
        h = np.maximum(h, 0).astype(np.float32)
        i += 1
    a = conv(h, "clf_base.")                                                   # :432
    t = h @ _f32(sd["clf_transformer.0.weight"]).T + sd["clf_transformer.0.bias"]
    t = np.maximum(bn(t, "clf_transformer.1."), 0).astype(np.float32)
    t = (t @ _f32(sd["clf_transformer.3.weight"]).T + sd["clf_transformer.3.bias"]).astype(np.float32)
    b = conv(t, "clf_target.")                                                 # :433
    c = conv(h, "clf_target.")                                                 # :434
    out = (O.log_softmax(a), O.log_softmax(c), O.log_softmax(b))               # :435 (base, target, target_hat)
    return out + (h,) if return_emb else out
