"""Seeded synthetic bridged graphs / embeddings for the BASELINE.json configs.

The reference ships no generator for its synthetic sets (only prose, `README.md:18`; paths are
commented out at `datasets.py:50-55`), so the shapes below are build-defined and recorded here.
Everything is numpy (PCG64, fixed seeds) so the build container, the CPU oracle and the GPU box
regenerate identical inputs; only arrays are produced, callers move them to the device.
"""
import numpy as np

__all__ = ["bridged_graph", "sync_rd_intra", "twitter_standin", "gaussian_embeddings", "random_multigraph", "edge_hash"]


def _local_or_uniform(rng, dst_rel, lo_domain, n_domain, cluster, p_local):
    """For each destination (given by its index relative to its own domain) draw a source id from
    domain [lo_domain, lo_domain+n_domain): same-position cluster w.p. p_local, else uniform."""
    n = dst_rel.shape[0]
    # map the destination's relative position onto the source domain (domains may differ in size)
    base = (dst_rel // cluster) * cluster
    base = np.minimum(base, max(n_domain - cluster, 0))
    local = base + rng.integers(0, min(cluster, n_domain), size=n)
    uni = rng.integers(0, n_domain, size=n)
    pick = rng.random(n) < p_local
    return lo_domain + np.where(pick, np.minimum(local, n_domain - 1), uni)


def bridged_graph(n_src, n_tar, k_within=6, k_cross=20, n_extra=0, cluster=1024, p_local=0.9, seed=0):
    """C4-style bridged graph.  Node order [sources ; targets] as `merge_graphs` produces
    (main_bridged_graph.py:163-193).  Edges are (from=neighbour, to=node), directed:
    k_within same-domain in-neighbours per node, k_cross source in-neighbours per target node
    (the kNN bridge, s->t only as in the shipped office graphs), n_extra further intra-domain
    edges.  Returns (edge_index int64 [2,E], central_mask bool [N])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    N = n_src + n_tar
    src_parts, dst_parts = [], []
    for lo, n in ((0, n_src), (n_src, n_tar)):
        if n == 0 or k_within == 0:
            continue
        dst_rel = np.repeat(np.arange(n, dtype=np.int64), k_within)
        src_parts.append(_local_or_uniform(rng, dst_rel, lo, n, cluster, p_local))
        dst_parts.append(lo + dst_rel)
    if k_cross > 0 and n_src > 0 and n_tar > 0:
        dst_rel = np.repeat(np.arange(n_tar, dtype=np.int64), k_cross)
        # relative position of the target inside T, rescaled onto S
        pos = (dst_rel * n_src) // n_tar
        src_parts.append(_local_or_uniform(rng, pos, 0, n_src, cluster, p_local))
        dst_parts.append(n_src + dst_rel)
    if n_extra > 0:
        d = rng.integers(0, N, size=n_extra)
        in_src = d < n_src
        rel = np.where(in_src, d, d - n_src)
        s_a = _local_or_uniform(rng, rel, 0, max(n_src, 1), cluster, p_local)
        s_b = _local_or_uniform(rng, rel, n_src, max(n_tar, 1), cluster, p_local)
        src_parts.append(np.where(in_src, s_a, s_b))
        dst_parts.append(d)
    ei = np.stack([np.concatenate(src_parts), np.concatenate(dst_parts)]).astype(np.int64)
    mask = np.zeros(N, dtype=bool)
    mask[:n_src] = True
    return ei, mask


def sync_rd_intra(n=10000, feat=64, homophily=0.7, deg=10, k_cross=20, seed=0):
    """C2: 'Sync-RD_intra' stand-in: n/2 source then n/2 target nodes, 2 classes,
    class means +-1 along random unit directions, target domain shifted by 0.5 and scaled by 1.5;
    `deg` intra-domain in-edges per node whose partner has the same label w.p. `homophily`;
    k_cross bridge edges per target by cosine on raw features; made undirected by the caller.
    Returns (x fp32 [n,feat], edge_index int64 [2,E], y int64 [n], central_mask bool [n])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ns = n // 2
    nt = n - ns
    y = rng.integers(0, 2, size=n)
    dirs = rng.standard_normal((2, feat))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    x = rng.standard_normal((n, feat)) + np.where(y[:, None] == 1, dirs[0], -dirs[1])
    x[ns:] = 1.5 * x[ns:] + 0.5
    x = x.astype(np.float32)
    src_parts, dst_parts = [], []
    for lo, m in ((0, ns), (ns, nt)):
        dst = np.repeat(np.arange(lo, lo + m), deg)
        same = rng.random(dst.shape[0]) < homophily
        want = np.where(same, y[dst], 1 - y[dst])
        ids = [lo + np.nonzero(y[lo:lo + m] == c)[0] for c in (0, 1)]
        pick = np.empty_like(dst)
        for c in (0, 1):
            sel = want == c
            pick[sel] = ids[c][rng.integers(0, len(ids[c]), size=int(sel.sum()))]
        src_parts.append(pick)
        dst_parts.append(dst)
    xn = x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-8)
    sim = xn[ns:] @ xn[:ns].T
    top = np.argsort(-sim, axis=1, kind="stable")[:, :k_cross]
    src_parts.append(top.reshape(-1))
    dst_parts.append(np.repeat(np.arange(ns, n), k_cross))
    ei = np.stack([np.concatenate(src_parts), np.concatenate(dst_parts)]).astype(np.int64)
    mask = np.zeros(n, dtype=bool)
    mask[:ns] = True
    return x, ei, y.astype(np.int64), mask


def _cosine_topk_f64(x_query, x_cand, k, chunk=2048):
    """Exact cosine top-k on the host in float64 (rank: score desc, index asc).  float64 keeps the selected SETS the same
    on every BLAS / CPU (neighbouring scores of N(0,1) features differ by >> 1e-15), which float32 GEMMs would not."""
    qn = x_query.astype(np.float64)
    qn /= np.maximum(np.linalg.norm(qn, axis=1, keepdims=True), 1e-12)
    cn = x_cand.astype(np.float64)
    cn /= np.maximum(np.linalg.norm(cn, axis=1, keepdims=True), 1e-12)
    import torch                                   # CPU torch: threaded float64 GEMM + top-k (numpy's argpartition took 50 s)
    out = np.empty((qn.shape[0], k), dtype=np.int64)
    cnt = torch.from_numpy(np.ascontiguousarray(cn.T))
    for s in range(0, qn.shape[0], chunk):
        sim = torch.from_numpy(qn[s:s + chunk]) @ cnt
        ps, part = torch.topk(sim, k, dim=1)
        ps, part = ps.numpy(), part.numpy()
        order = np.lexsort((part, -ps), axis=1)
        out[s:s + chunk] = np.take_along_axis(part, order, axis=1)
    return out


def twitter_standin(n_src=581, n_tar=20230, feat=300, n_random=450_000, k_within=6, k_cross=20, seed=0):
    """C3 stand-in for the absent Twitter_Graph bridged graph (shape from the reference's loader and recipe:
    F=300 `dataset_ktgnn.py:81-82`, k_within=6 / k_cross=20 / hidden 128 / `--to_undirected` `run.sh:5-7`, original edges
    undirected `datasets.py:24-29`; the node / edge counts are the survey's unverified shape hint).  Node order
    [sources ; targets]; features i.i.d. N(0,1); `n_random` directed random edges over ALL nodes (intra + inter domain;
    ~0.9 M once undirected) + cosine kNN bridge edges on the raw features (from = neighbour, to = query): `k_within`
    same-domain neighbours per node (self matches kept, as in the reference, Appendix B-3) and `k_cross` source
    neighbours per target node.  The caller makes it undirected (`ToUndirected`).
    Returns (x fp32 [N,feat], edge_index int64 [2,E], y int64 [N], central_mask bool [N])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = n_src + n_tar
    x = rng.standard_normal((n, feat), dtype=np.float32)
    y = rng.integers(0, 2, size=n).astype(np.int64)
    src_parts = [rng.integers(0, n, size=n_random)]
    dst_parts = [rng.integers(0, n, size=n_random)]
    for lo, m in ((0, n_src), (n_src, n_tar)):
        if k_within > 0 and m > 0:
            top = _cosine_topk_f64(x[lo:lo + m], x[lo:lo + m], min(k_within, m))
            src_parts.append(lo + top.reshape(-1))
            dst_parts.append(np.repeat(np.arange(lo, lo + m, dtype=np.int64), top.shape[1]))
    if k_cross > 0 and n_src > 0 and n_tar > 0:
        top = _cosine_topk_f64(x[n_src:], x[:n_src], min(k_cross, n_src))
        src_parts.append(top.reshape(-1))
        dst_parts.append(np.repeat(np.arange(n_src, n, dtype=np.int64), top.shape[1]))
    ei = np.stack([np.concatenate(src_parts), np.concatenate(dst_parts)]).astype(np.int64)
    mask = np.zeros(n, dtype=bool)
    mask[:n_src] = True
    return x, ei, y, mask


def edge_hash(edge_index, num_nodes):
    """Order-independent 61-bit checksum of an edge multiset (fixtures store it so a regenerated graph can be checked)."""
    key = edge_index[0].astype(np.uint64) * np.uint64(num_nodes) + edge_index[1].astype(np.uint64)
    key = (key ^ (key >> np.uint64(29))) * np.uint64(0x9E3779B97F4A7C15)      # wraps mod 2^64
    return int(np.bitwise_xor.reduce(key) & np.uint64((1 << 61) - 1)) ^ int(edge_index.shape[1])


def gaussian_embeddings(n, d=128, seed=0):
    """C5: i.i.d. N(0,1) fp32 embeddings [n,d] (d = `Similar.lin_self` output width, models.py:98)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal((n, d), dtype=np.float32)


def random_multigraph(n, e, frac_src=0.5, n_isolated=0, seed=0):
    """Small adversarial test graph: uniform random edges incl. self loops and duplicate edges;
    the last `n_isolated` nodes receive no in-edges (only the rewritten self loop) and an arbitrary
    (non-contiguous) domain mask.  Returns (edge_index int64 [2,e], central_mask bool [n])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    src = rng.integers(0, n, size=e)
    dst = rng.integers(0, max(n - n_isolated, 1), size=e)
    mask = rng.random(n) < frac_src
    if mask.all() or (~mask).all():
        mask[0], mask[-1] = True, False
    return np.stack([src, dst]).astype(np.int64), mask
