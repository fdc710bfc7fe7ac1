"""Bridged-graph (kNN) construction on MI355X -- host-side mirror of the reference's step-1 hot
functions (Bridged-GNN/main_bridged_graph.py:33-120, :163-193) over the HIP library.

The reference enumerates every (candidate, query) index pair per batch, re-runs both encoders and
pushes [Ns*B, 128] gathers through the scorer.  Here the per-node part of each
scorer is evaluated once per node and the per-pair part is a streaming top-k kernel:
  cosine (`Similar`, `Similar_v2(mode='cosine')`, `Similar_noTrans`; models/models.py:124-130,
          :945-948, :185-189):  q = u + biasatt(u), u = lin_self(z)  per node, then
          normalise rows and run `bgnn_cosine_topk_f32` (bf16-piece MFMA shortlist pass + canonical fp64 re-score);
  mlp    (`Similar_v2(mode='mlp')`, :949-951): eval-mode BN/Linear are affine, so
          W1.bn1([z_s||z_t]) = A[s] + B[t]; `bgnn_mlp_pair_topk_f32` evaluates
          w2.relu(bn2(A[s]+B[t])) + b2 per pair.
`batch_size`/`epsilon` are accepted for signature compatibility (epsilon is unused by the reference
too, main_bridged_graph.py:33); batching is internal to the kernels.
"""
import torch
import torch.nn.functional as F

from . import ops
from .data import Data

__all__ = ["BridgeScorer", "add_topk_sim_cross_domain_edges", "add_topk_sim_within_domain_edges", "sharded_cosine_topk_edges",
           "merge_graphs", "pair_enumeration", "check_added_edges_cross_domain_validity",
           "check_added_edges_within_domain_validity", "align_e_sim_to_edges", "reorder", "eval_bridged_Graph",
           "eval_homophily", "gen_bridged_graph"]

_BN_EPS = 1e-5


def pair_enumeration(x1, x2):
    """models/models.py:265-282 (x2-major enumeration).  Provided for API parity / small tests only:
    the top-k path never materialises pairs."""
    assert x1.dim() == 2 and x2.dim() == 2, "Input dimension must be 2"
    x1_ = x1.repeat(x2.size(0), 1)
    x2_ = x2.repeat(1, x1.size(0)).view(-1, x1.size(1))
    return torch.cat((x1_, x2_), dim=1)


def _bn_affine(sd, prefix):
    inv = sd[prefix + "weight"] / torch.sqrt(sd[prefix + "running_var"] + _BN_EPS)
    return inv, sd[prefix + "bias"] - sd[prefix + "running_mean"] * inv


class BridgeScorer:
    """Eval-mode pair scorer + (v2/mlp-backbone) encoders of the reference's trained similarity
    learner, built from its checkpoint state_dict (`Adversarial_Learner{,_v2}`,
    models/models.py:815-844, :1110-1142; key layout in SURVEY.md 8(b)).  Weights live on `device`."""

    def __init__(self, state_dict, device, norm_mode="None", norm_scale=1.0):
        sd = {k: v.to(device=device, dtype=torch.float32) if v.is_floating_point() else v.to(device)
              for k, v in state_dict.items()}
        self.sd, self.device = sd, torch.device(device)
        self.norm_mode, self.norm_scale = norm_mode, norm_scale
        p = "source_learner.sim_net."
        self.sim = {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
        self.version = "v2" if "source_learner.backbone.layers.0.weight" in sd else "v1"
        if "biasatt.0.weight" in self.sim:
            self.sim_mode = "cosine"
        elif "lin_self.4.bias" in self.sim:
            self.sim_mode = "mlp"
        else:
            self.sim_mode = "cosine_notrans"        # Similar_noTrans: q = z
        self.use_clf = "lin_clf.weight" in self.sim

    # ---- encoders (dense per-node work, run ONCE; reference re-runs them per batch :56,:100) ----
    def _pairnorm(self, x):
        """models/models.py:29-64"""
        mode, scale = self.norm_mode, self.norm_scale
        if mode == "None":
            return x
        col_mean = x.mean(dim=0)
        if mode == "PN":
            x = x - col_mean
            return scale * x / (1e-6 + x.pow(2).sum(dim=1).mean()).sqrt()
        if mode == "PN-SI":
            x = x - col_mean
            return scale * x / (1e-6 + x.pow(2).sum(dim=1, keepdim=True)).sqrt()
        if mode == "PN-SCS":
            return scale * x / (1e-6 + x.pow(2).sum(dim=1, keepdim=True)).sqrt() - col_mean
        raise NotImplementedError(mode)

    def _mlp_encoder(self, x, prefix):
        """`MLP.forward` eval mode, models/models.py:880-893 (use_norm=True at every v2 call site)."""
        sd = self.sd
        x = F.linear(x, sd[prefix + "layers.0.weight"], sd[prefix + "layers.0.bias"])
        x = F.relu(self._pairnorm(x))
        return F.linear(x, sd[prefix + "layers.1.weight"], sd[prefix + "layers.1.bias"])

    def _sage_encoder(self, x, edge_index, prefix):
        """`GraphEncoder.forward` eval mode (2 x SAGEConv, mean aggregation, root weight), models/models.py:220-263
        (SURVEY.md 8(f) rank 3).  lin_l is linear, so W_l.mean_j(x_j) = mean_j(W_l x_j): transform first, then the
        fused CSR kernel averages the out-width rows (uniform attention == mean; rows without in-edges give 0)."""
        sd, dev = self.sd, self.device
        n = x.shape[0]
        csr = ops.build_dst_csr(edge_index.to(dev).long().contiguous(), n, rewrite_self_loops=False)
        ones = torch.ones(n, dtype=torch.uint8, device=dev)
        n_layers = 1 + max(int(k[len(prefix) + 6:].split(".")[0]) for k in sd if k.startswith(prefix + "convs."))
        for i in range(n_layers):
            pl = f"{prefix}convs.{i}."
            D = sd[pl + "lin_l.weight"].shape[0]
            xw = F.linear(x, sd[pl + "lin_l.weight"])                                         # [n, D]
            tab = torch.zeros(n, ops.pad4(D), dtype=torch.float32, device=dev)
            tab[:, :D] = xw
            zero_a = torch.zeros(D, dtype=torch.float32, device=dev)
            agg = ops.adaptedconv_aggregate(tab, tab, zero_a, zero_a, csr, ones, D)[:, :D]     # mean over in-edges
            out = agg + sd[pl + "lin_l.bias"]
            if pl + "lin_r.weight" in sd:
                out = out + F.linear(x, sd[pl + "lin_r.weight"])                              # root weight
            x = out if i == n_layers - 1 else F.relu(self._pairnorm(out))                     # :249-259
        return x

    def encode_source(self, data):
        """z_src = source_learner.backbone(x, edge_index) (models.py:835 / :1133)."""
        x = data.x.to(self.device).float()
        if self.version != "v2":
            return self._sage_encoder(x, data.edge_index, "source_learner.backbone.")
        return self._mlp_encoder(x, "source_learner.backbone.")

    def encode_target(self, data):
        """z_tar, _ = target_learner.encode(data) (models.py:1092-1096)."""
        sd = self.sd
        h0 = F.linear(data.x.to(self.device).float(), sd["target_learner.equavilent_trans_layer.0.weight"],
                      sd["target_learner.equavilent_trans_layer.0.bias"])
        h0 = torch.tanh(self._pairnorm(h0))
        if self.version != "v2":
            return self._sage_encoder(h0, data.edge_index, "target_learner.encoder.")
        return self._mlp_encoder(h0, "target_learner.encoder.")

    def class_probs(self, z):
        """exp(log_softmax(lin_clf(relu(z)))) -- models.py:136-140, :842 (dropout is identity in eval)."""
        if not self.use_clf:
            return None
        return F.softmax(F.linear(F.relu(z), self.sim["lin_clf.weight"], self.sim["lin_clf.bias"]), dim=-1)

    # ---- per-node halves of the pair scorers ---------------------------------------------------
    def cosine_q(self, z):
        """q = u + biasatt(u), u = lin_self(z), eval-mode BN (models.py:93-99, :70-74, :125-127)."""
        if self.sim_mode == "cosine_notrans":
            return z.contiguous()
        s = self.sim
        a0, c0 = _bn_affine(s, "lin_self.0.")
        u = F.linear(z * a0 + c0, s["lin_self.1.weight"])
        a2, c2 = _bn_affine(s, "lin_self.2.")
        u = F.linear(torch.tanh(u * a2 + c2), s["lin_self.4.weight"])
        b = F.linear(torch.tanh(F.linear(u, s["biasatt.0.weight"], s["biasatt.0.bias"])),
                     s["biasatt.2.weight"], s["biasatt.2.bias"])
        return (u + b).contiguous()

    def mlp_consts(self):
        """bn2 scale / shift, w2, b2 of the mlp scorer (models.py:921-925); constants of the checkpoint, formed once (the
        bias read-back is the only host sync of the mlp path)."""
        c = getattr(self, "_mlp_consts", None)
        if c is None:
            s = self.sim
            scale, shift = _bn_affine(s, "lin_self.2.")
            c = self._mlp_consts = (scale.contiguous(), shift.contiguous(), s["lin_self.4.weight"].reshape(-1).contiguous(),
                                    float(s["lin_self.4.bias"].reshape(-1)[0].item()))
        return c

    def mlp_term_cand(self, z_cand):
        """A[cand]: the idx1 ('from' / candidate) half of the first Linear on the concatenation (models.py:918-920)"""
        s = self.sim
        h = z_cand.shape[1]
        a, c = _bn_affine(s, "lin_self.0.")
        return F.linear(z_cand * a[:h] + c[:h], s["lin_self.1.weight"][:, :h]).contiguous()

    def mlp_term_query(self, z_query):
        """B[query]: the idx2 half, carrying the Linear's bias"""
        s = self.sim
        h = z_query.shape[1]
        a, c = _bn_affine(s, "lin_self.0.")
        return F.linear(z_query * a[h:] + c[h:], s["lin_self.1.weight"][:, h:], s["lin_self.1.bias"]).contiguous()

    def mlp_terms(self, z_cand, z_query):
        """A[cand], B[query], bn2 scale/shift, w2, b2 (models.py:918-925, :949-951); first half of the
        concatenation is the idx1 ('from'/candidate) side."""
        return (self.mlp_term_cand(z_cand), self.mlp_term_query(z_query)) + self.mlp_consts()

    def topk(self, z_cand, z_query, k, rank=0, world=1, group=None):
        """-> (idx [Nq,k] int64, probs [Nq,k] fp32, n_fallback) ; rows sorted by (score desc, idx asc).
        world > 1: this rank scores the query rows `shard_range(Nq, rank, world)` only (-> [Nq/world, k]
        tables); the per-node halves of the scorer are evaluated on the rank's slice of the candidate rows and made whole
        by ONE all_gather; no cross-rank merge of top-k lists is needed under row partitioning."""
        from .dist import all_gather_rows, shard_range
        same = z_query is z_cand
        clo, chi = shard_range(z_cand.shape[0], rank, world)
        qlo, qhi = shard_range(z_query.shape[0], rank, world)
        zc, zq = z_cand[clo:chi], z_query[qlo:qhi]
        if self.sim_mode == "mlp":
            A, B = self.mlp_term_cand(zc), self.mlp_term_query(zq)
            scale, shift, w2, b2 = self.mlp_consts()
            A = all_gather_rows(A, group, world)
            return ops.mlp_pair_topk(A.contiguous(), B, scale, shift, w2, b2, k, apply_sigmoid=True)
        qc_local = ops.l2_normalize_rows(self.cosine_q(zc))
        qc = all_gather_rows(qc_local, group, world)
        qq = qc[qlo:qhi] if same else ops.l2_normalize_rows(self.cosine_q(zq))
        return ops.cosine_topk(qq.contiguous(), qc.contiguous(), k, apply_sigmoid=True)

    # ---- reference-shaped entry points (explicit pair lists; small inputs / API parity) ---------
    def pair_probs(self, z1, z2, idx1, idx2):
        if self.sim_mode == "mlp":
            A, B, scale, shift, w2, b2 = self.mlp_terms(z1, z2)
            t = F.relu((A[idx1] + B[idx2]) * scale + shift)
            return torch.sigmoid(t @ w2 + b2)
        q1, q2 = self.cosine_q(z1), self.cosine_q(z2)
        return torch.sigmoid(F.cosine_similarity(q1[idx1], q2[idx2], dim=1, eps=1e-8))

    def get_probs_cross_domain(self, data_src, data_tar, idx1, idx2, return_representation=False):
        """models/models.py:834-844 / :1132-1142"""
        z_src, z_tar = self.encode_source(data_src), self.encode_target(data_tar)
        probs = self.pair_probs(z_src, z_tar, idx1, idx2).unsqueeze(-1)
        out = (probs, self.class_probs(z_src), self.class_probs(z_tar))
        return out + (z_src, z_tar) if return_representation else out

    def get_probs_within_domain(self, data, idx1, idx2, domain="target"):
        """models/models.py:824-833 / :1122-1131"""
        z = self.encode_source(data) if domain == "source" else self.encode_target(data)
        return self.pair_probs(z, z, idx1, idx2).unsqueeze(-1), self.class_probs(z)


def _homophily(y_from, y_to, edge_index):
    lab = (y_from[edge_index[0]] != -1) & (y_to[edge_index[1]] != -1)
    same = (y_from[edge_index[0]] == y_to[edge_index[1]]) & lab
    return (same.sum() / lab.sum()).item() if int(lab.sum()) > 0 else float("nan")


def sharded_cosine_topk_edges(q_local, cand_local, k, rank=0, world=1, group=None, query_base=0, events=None,
                              apply_sigmoid=True, backend=None):
    """The cosine kNN bridge on raw embeddings, sharded by query rows (main_bridged_graph.py:45-68 is a loop over query
    batches -- the natural shard).  `q_local` = this rank's query rows (global ids start at `query_base`),
    `cand_local` = this rank's slice of the candidate rows: each rank normalises its slice, ONE all_gather makes the
    candidates whole (rank order = global candidate id order), every rank scores its queries against all of them.
    -> (coalesced edge_index of this rank's queries with GLOBAL ids, idx [nq,k], probs [nq,k], n_fallback).
    The union of the ranks' edge lists, coalesced, is the single-rank edge list (tests/test_dist_gloo.py).
    `events` = (start, end) HIP events recorded around the top-k call.  `backend`: namespace providing l2_normalize_rows /
    cosine_topk / topk_edges / coalesce (default: the HIP ops; the CPU gloo test of this host logic injects the oracle)."""
    from .dist import all_gather_rows
    be = ops if backend is None else backend
    qn = be.l2_normalize_rows(q_local)
    cn = all_gather_rows(be.l2_normalize_rows(cand_local), group, world)
    if events is not None:
        events[0].record()
    idx, val, nfb = be.cosine_topk(qn, cn.contiguous(), k, apply_sigmoid=apply_sigmoid)
    if events is not None:
        events[1].record()
    fused = getattr(be, "topk_edges_coalesced", None)       # distinct valid candidates per query: coalescing = one stable pair sort
    if fused is not None and k <= cn.shape[0]:
        ei = fused(idx, cn.shape[0], cand_base=0, query_base=int(query_base))
    else:
        ei = be.coalesce(be.topk_edges(idx, cand_base=0, query_base=int(query_base)))
    return ei, idx, val, nfb


def gather_edges(edge_index, group=None, world=None, backend=None):
    """per-rank coalesced edge lists -> the global coalesced list on every rank (only when one `Data` must be
    materialised; partitioned by destination is the layout half B wants anyway)."""
    from .dist import all_gather_rows
    be = ops if backend is None else backend
    allv = all_gather_rows(edge_index.t().contiguous(), group, world).t().contiguous()
    return be.coalesce(allv)


def add_topk_sim_cross_domain_edges(data_src, data_tar, model, epsilon=0.5, k=3, batch_size=1000,
                                    z_src=None, z_tar=None, verbose=True, rank=0, world=1, group=None):
    """main_bridged_graph.py:33-75.  Returns (coalesced edge_index [2,E] (row0 = source id, row1 =
    target id), e_sim_mat [Nt,k] sigmoid probs in top-k order, idx_src_mat [Nt,k] int64,
    probs_clf_src [Ns,C], probs_clf_tar [Nt,C]); all CUDA tensors.
    world > 1: the target (query) rows are sharded over the ranks (`BridgeScorer.topk`), the [Nt/world, k] tables are
    all-gathered, and every rank returns the same 5-tuple as a single-rank call."""
    z_src = model.encode_source(data_src) if z_src is None else z_src.to(model.device).float()
    z_tar = model.encode_target(data_tar) if z_tar is None else z_tar.to(model.device).float()
    idx, probs, _ = model.topk(z_src.contiguous(), z_tar.contiguous(), k, rank=rank, world=world, group=group)
    if world > 1:
        from .dist import all_gather_rows
        idx, probs = all_gather_rows(idx, group, world), all_gather_rows(probs, group, world)
    # :61-68 + coalesce :75 -- the k candidates of a query are distinct, so the coalesced list is one stable pair sort
    n_src = z_src.shape[0]
    edge_index_added = ops.topk_edges_coalesced(idx, n_src) if k <= n_src else ops.coalesce(ops.topk_edges(idx))
    if verbose and hasattr(data_src, "y") and hasattr(data_tar, "y"):
        ys, yt = data_src.y.to(model.device), data_tar.y.to(model.device)
        print("Current homophily ratio:", _homophily(ys, yt, edge_index_added))  # :71-74 (a ratio over the edge SET)
    return (edge_index_added, probs, idx,                                      # :75
            model.class_probs(z_src), model.class_probs(z_tar))


def add_topk_sim_within_domain_edges(data_src, model, k=3, batch_size=1000, domain="source", z=None, verbose=True):
    """main_bridged_graph.py:77-120.  Candidates = all nodes of the domain INCLUDING the query itself
    (the reference does not remove self matches).  Returns (coalesced
    edge_index (from = top-k idx, to = query), e_sim_mat [n,k], idx_mat [n,k])."""
    if z is None:
        z = model.encode_source(data_src) if domain == "source" else model.encode_target(data_src)
    z = z.to(model.device).float().contiguous()
    idx, probs, _ = model.topk(z, z, k)
    ei = ops.topk_edges_coalesced(idx, z.shape[0]) if k <= z.shape[0] else ops.coalesce(ops.topk_edges(idx))   # :112-113
    if verbose and hasattr(data_src, "y"):
        y = data_src.y.to(model.device)
        print("Current homophily ratio of Graph:", _homophily(y, y, ei))       # :116-119
    return ei, probs, idx


def merge_graphs(data_src, data_tar, edge_index_cross_added, edge_index_added_src=None, edge_index_added_tar=None):
    """main_bridged_graph.py:163-193: node order [sources ; targets]; returns the coalesced bridged
    `Data` (x, edge_index, y, train/val/test/central masks) on the device of the edge tensors.
    Unlike the reference (:170) the cross-edge argument is NOT modified in place."""
    dev = edge_index_cross_added.device
    ns, nt = data_src.x.shape[0], data_tar.x.shape[0]
    n = ns + nt
    parts = [data_src.edge_index.to(dev), data_tar.edge_index.to(dev) + ns,
             torch.stack([edge_index_cross_added[0], edge_index_cross_added[1] + ns])]
    if edge_index_added_src is not None:
        parts.append(edge_index_added_src.to(dev))
    if edge_index_added_tar is not None:
        parts.append(edge_index_added_tar.to(dev) + ns)
    edge_index = ops.coalesce(torch.cat(parts, dim=1), n)                      # :193 Data(...).coalesce()
    central = torch.zeros(n, dtype=torch.bool, device=dev)
    central[:ns] = True
    ys, yt = data_src.y.to(dev), data_tar.y.to(dev)
    train = central.clone()
    train[:ns] &= ys != -1                                                     # :186-187
    val = torch.zeros(n, dtype=torch.bool, device=dev)
    test = torch.zeros(n, dtype=torch.bool, device=dev)
    if hasattr(data_tar, "train_mask"):
        train[ns:] = data_tar.train_mask.to(dev)                               # :188
    if hasattr(data_tar, "val_mask"):
        val[ns:] = data_tar.val_mask.to(dev)                                   # :189
    if hasattr(data_tar, "test_mask"):
        test[ns:] = data_tar.test_mask.to(dev)                                 # :190
    return Data(x=torch.cat((data_src.x.to(dev), data_tar.x.to(dev)), dim=0), edge_index=edge_index,
                y=torch.cat((ys, yt), dim=0), train_mask=train, val_mask=val, test_mask=test, central_mask=central)


# ------------------------------------------------------------------------------------------------
# Edge validity filters: main_bridged_graph.py:123-161 and :225-264.
# Cheap per-edge element-wise work between top-k and merge -> plain torch ops on the GPU tensors.
def align_e_sim_to_edges(edge_index, e_sim_mat, idx_mat):
    """Similarity of every COALESCED edge.  The reference passes `e_sim_mat.view(-1)` (top-k order) next to the
    coalesced (source-sorted) edge list, so its confidence filter hits the wrong edges;
    this looks each edge (from, to) up in the [query, k] tables instead."""
    k = idx_mat.shape[1]
    cand = idx_mat[edge_index[1]]                                  # [E, k] candidates of the edge's query
    hit = cand == edge_index[0].unsqueeze(1)
    pos = hit.float().argmax(dim=1)
    assert bool(hit.any(dim=1).all()), "edge list does not come from these top-k tables"
    return e_sim_mat[edge_index[1], pos]


def _filter_report(verbose, *a):
    if verbose:
        print(*a)


def check_added_edges_cross_domain_validity(edge_index_added, e_sim, data_src, data_tar, probs_clf_src, probs_clf_tar,
                                            thres_conf_quantile=0.1, thres_feat_sim=0.0, verbose=False):
    """main_bridged_graph.py:225-264.  `e_sim`: one similarity per edge.  Pass `align_e_sim_to_edges(...)` for the
    intended semantics, or the reference's own `e_sim_mat.view(-1)` to reproduce its index-misaligned behaviour
    bit for bit (same length, different order)."""
    dev = edge_index_added.device
    e0, e1 = edge_index_added[0], edge_index_added[1]
    ys, yt = data_src.y.to(dev), data_tar.y.to(dev)
    pred_s, pred_t = probs_clf_src.argmax(dim=1), probs_clf_tar.argmax(dim=1)
    e_sim = e_sim.reshape(-1).to(dev)
    rm = torch.zeros(edge_index_added.shape[1], dtype=torch.bool, device=dev)
    thres = e_sim.quantile(q=thres_conf_quantile)                                  # :238
    rm |= e_sim < thres                                                            # :239
    _filter_report(verbose, "1. low SimNet confidence:", int(rm.sum()))
    rm |= pred_s[e0] != ys[e0]                                                     # :243
    rm |= (pred_t[e1] != yt[e1]) & data_tar.train_mask.to(dev)[e1]                 # :244
    rm |= pred_s[e0] != pred_t[e1]                                                 # :248
    cos = F.cosine_similarity(data_src.x.to(dev)[e0], data_tar.x.to(dev)[e1])      # :252
    rm |= cos < thres_feat_sim                                                     # :253
    _filter_report(verbose, "[Done] removed", int(rm.sum()), "of", rm.numel())
    return edge_index_added[:, ~rm]                                                # :257


def check_added_edges_within_domain_validity(edge_index_added, e_sim, data_in, probs_clf, thres_conf_quantile=0.1,
                                             thres_feat_sim=0.0, verbose=False):
    """main_bridged_graph.py:123-161 (same remark about `e_sim` alignment)."""
    dev = edge_index_added.device
    e0, e1 = edge_index_added[0], edge_index_added[1]
    y = data_in.y.to(dev)
    pred = probs_clf.argmax(dim=1)
    e_sim = e_sim.reshape(-1).to(dev)
    rm = torch.zeros(edge_index_added.shape[1], dtype=torch.bool, device=dev)
    rm |= e_sim < e_sim.quantile(q=thres_conf_quantile)                            # :135-136
    tm = data_in.train_mask.to(dev)[e1]
    rm |= (pred[e0] != y[e0]) & tm                                                 # :140
    rm |= (pred[e1] != y[e1]) & tm                                                 # :141
    rm |= pred[e0] != pred[e1]                                                     # :145
    cos = F.cosine_similarity(data_in.x.to(dev)[e0], data_in.x.to(dev)[e1])        # :149
    rm |= cos < thres_feat_sim                                                     # :150
    _filter_report(verbose, "[Done] removed", int(rm.sum()), "of", rm.numel())
    return edge_index_added[:, ~rm]                                                # :154


# ------------------------------------------------------------------------------------------------
# Graph assembly tail of step 1: reorder / diagnostics / the gen_bridged_graph pipeline.
def _as_index(mapper, n):
    """orig-id -> local-index mapping given as a dict (reference `dataset_conversion`, utils.py:58-63) or as a
    LongTensor of the original ids in local order; returns the LongTensor form."""
    if hasattr(mapper, "orig"):                      # IndexMapper of this package's dataset_conversion
        return mapper.orig
    if isinstance(mapper, dict):
        out = torch.empty(n, dtype=torch.int64)
        for orig, loc in mapper.items():
            out[loc] = orig
        return out
    return torch.as_tensor(mapper, dtype=torch.int64)


def reorder(data_merge, data_src, mapper_idx_src, mapper_idx_tar):
    """main_bridged_graph.py:195-222 without the per-edge Python loop (:221): put the merged graph back into the
    node order of the original dataset.  Original ids must be a permutation of 0..N-1 (as in the reference)."""
    n_src = data_src.x.shape[0] if hasattr(data_src, "x") else int(data_src)
    dev = data_merge.x.device
    n = data_merge.x.shape[0]
    orig_of_merged = torch.cat([_as_index(mapper_idx_src, n_src), _as_index(mapper_idx_tar, n - n_src)]).to(dev)
    assert bool((torch.sort(orig_of_merged).values == torch.arange(n, device=dev)).all()), "ids must be a permutation"
    merged_of_orig = torch.empty_like(orig_of_merged)
    merged_of_orig[orig_of_merged] = torch.arange(n, device=dev)                  # reorder_idxs (:209)
    for key in ("train_mask", "val_mask", "test_mask", "central_mask", "x", "y"):   # :211-216
        if hasattr(data_merge, key):
            setattr(data_merge, key, getattr(data_merge, key)[merged_of_orig])
    data_merge.edge_index = orig_of_merged[data_merge.edge_index]                 # :221
    return data_merge


def _dataset_split(data, num_classes, ratio):
    """utils.py:20-38.  The permutations come from torch's global CPU generator in the reference's call order (one `randperm` per
    class), so a `set_random_seed(seed)` in front reproduces its split bit for bit; everything else is index arithmetic."""
    import numpy as np
    y_cpu = data.y.cpu()
    for c in range(num_classes):
        idx = (y_cpu == c).nonzero(as_tuple=False).view(-1)
        nc = idx.shape[0]
        n_train = int(np.ceil(nc * ratio[0]))
        n_val = int(np.floor(nc * ratio[1]))
        assert nc - n_train - n_val >= 0
        perm = torch.randperm(nc)
        dev = data.train_mask.device
        data.train_mask[idx[perm[:n_train]].to(dev)] = True
        data.val_mask[idx[perm[n_train:n_train + n_val]].to(dev)] = True
        data.test_mask[idx[perm[n_train + n_val:]].to(dev)] = True


class IndexMapper(dict):
    """orig-id -> local-index mapping of `dataset_conversion` (the reference returns a Python dict built in a per-node loop,
    utils.py:58-63).  Still a dict for the reference's call sites (`reorder`, main_bridged_graph.py:199-203), built from and
    carrying the LongTensor `orig` of original ids in local order, which this package's vectorised `reorder` uses directly."""

    def __init__(self, orig):
        orig = torch.as_tensor(orig, dtype=torch.int64).cpu()
        super().__init__(zip(orig.tolist(), range(orig.shape[0])))
        self.orig = orig


def dataset_conversion(data, seed=0, train_val_test_ratio=(0.6, 0.2, 0.2), dataset_name=None, split_data=True):
    """utils.py:41-99: cut a VS-graph into its source (`central_mask`) and target component with local node ids, labels and
    fresh train / val / test masks -> (data_src, data_tar, mapper_idx_src, mapper_idx_tar).  Same results as the reference
    (fixture `tests/golden/f4_utils.npz`, both split modes) without its per-node / per-edge Python loops: the id maps are one
    `cumsum`, the edge relabelling one gather.  Works on the device `data` lives on; the class permutations are drawn from
    torch's CPU generator in the reference's order."""
    from .data import Data
    from .utils import set_random_seed
    set_random_seed(seed)
    cm = data.central_mask.bool()
    dev = cm.device
    x_src, x_tar = data.x[cm], data.x[~cm]
    if dataset_name in ("company", "twitter"):                                   # :45-49
        x_tar = x_tar[:, :33 if dataset_name == "company" else 300]
    idx_src, idx_tar = torch.where(cm)[0], torch.where(~cm)[0]
    local = torch.where(cm, torch.cumsum(cm.long(), 0) - 1, torch.cumsum((~cm).long(), 0) - 1)   # id inside the node's own domain
    ei = data.edge_index
    s_src, s_dst = cm[ei[0]], cm[ei[1]]
    ei_src = local[ei[:, s_src & s_dst]]                                          # :66,:68 (edge order kept)
    ei_tar = local[ei[:, ~s_src & ~s_dst]]

    def blank(n):
        return torch.zeros(n, dtype=torch.bool, device=dev)
    ns, nt = idx_src.shape[0], idx_tar.shape[0]
    data_src = Data(x=x_src, edge_index=ei_src, y=data.y[cm], train_mask=blank(ns), val_mask=blank(ns), test_mask=blank(ns))
    data_tar = Data(x=x_tar, edge_index=ei_tar, y=data.y[~cm], train_mask=blank(nt), val_mask=blank(nt), test_mask=blank(nt))
    num_classes = int(data.y.max().item()) + 1
    _dataset_split(data_src, num_classes, train_val_test_ratio)                   # :81
    if split_data:
        _dataset_split(data_tar, num_classes, train_val_test_ratio)               # :83
    else:                                                                         # :85-95: keep the data's own split
        for key in ("train_mask", "val_mask", "test_mask"):
            setattr(data_tar, key, getattr(data, key).bool()[~cm].clone())
    return data_src, data_tar, IndexMapper(idx_src), IndexMapper(idx_tar)


def eval_bridged_Graph(data_merge, verbose=False):
    """utils.py:101-113: share of test nodes whose labelled in-neighbourhood is majority same-label."""
    y, ei = data_merge.y, data_merge.edge_index
    n = y.shape[0]
    C = int(y.max().item()) + 1
    onehot = F.one_hot(y + 1, C + 1).float()[:, 1:]                               # unlabeled (-1) -> zero row
    lbl = torch.zeros(n, C, device=y.device).index_add_(0, ei[1], onehot[ei[0]])  # adj_t @ y_onehot
    deg = lbl.sum(1)
    ok = (deg != 0) & (y != -1)
    deg = torch.where(ok, deg, deg + 1e-3)
    local = (lbl * onehot).sum(1) / deg
    tm = data_merge.test_mask
    ratio = (local[tm] > 0.5).sum() / tm.sum()
    if verbose:
        print(ratio)
    return ratio


def eval_homophily(data, second_order=False, verbose=False):
    """utils.py:115-131: labelled-edge homophily; the 2-hop variant uses a sparse product instead of the reference's
    dense N x N matrix (:121) and is therefore optional."""
    y, ei = data.y, data.edge_index

    def ratio(e):
        lab = (y[e[0]] != -1) & (y[e[1]] != -1)
        return ((y[e[0]] == y[e[1]]) & lab).sum() / lab.sum()
    r1 = ratio(ei)
    r2 = None
    if second_order:
        n = y.shape[0]
        A = torch.sparse_coo_tensor(ei, torch.ones(ei.shape[1], device=ei.device), (n, n)).coalesce()
        r2 = ratio(torch.sparse.mm(A, A).coalesce().indices())
    if verbose:
        print("homophily ratio:", float(r1), "" if r2 is None else f"2nd: {float(r2)}")
    return (r1, r2) if second_order else r1


def gen_bridged_graph(data_src, data_tar, model, k_cross=20, k_within=6, check_cross=False, check_within=False,
                      thres_conf_quantile=0.1, thres_feat_sim=0.8, mapper_idx_src=None, mapper_idx_tar=None,
                      save_path=None, reference_filter_quirk=False, verbose=False):
    """main_bridged_graph.py:267-321 after the model is loaded: top-k cross edges (+ filter), top-k within edges per
    domain (+ filter; the reference hard-codes quantile 0.1 / feature threshold 0.8 there, :302-306), merge, reorder,
    save.  `reference_filter_quirk=True` feeds the filters the top-k-ordered similarity vector like the reference."""
    z_src, z_tar = model.encode_source(data_src), model.encode_target(data_tar)
    ec, esim, eidx, pcs, pct = add_topk_sim_cross_domain_edges(data_src, data_tar, model, k=k_cross, z_src=z_src, z_tar=z_tar,
                                                               verbose=verbose)
    if check_cross:
        es = esim.reshape(-1) if reference_filter_quirk else align_e_sim_to_edges(ec, esim, eidx)
        ec = check_added_edges_cross_domain_validity(ec, es, data_src, data_tar, pcs, pct, thres_conf_quantile,
                                                     thres_feat_sim, verbose)
    e_s = e_t = None
    if k_within > 0:
        e_s, sim_s, idx_s = add_topk_sim_within_domain_edges(data_src, model, k=k_within, domain="source", z=z_src, verbose=verbose)
        e_t, sim_t, idx_t = add_topk_sim_within_domain_edges(data_tar, model, k=k_within, domain="target", z=z_tar, verbose=verbose)
        if check_within:
            a = sim_s.reshape(-1) if reference_filter_quirk else align_e_sim_to_edges(e_s, sim_s, idx_s)
            b = sim_t.reshape(-1) if reference_filter_quirk else align_e_sim_to_edges(e_t, sim_t, idx_t)
            e_s = check_added_edges_within_domain_validity(e_s, a, data_src, pcs, 0.1, 0.8, verbose)
            e_t = check_added_edges_within_domain_validity(e_t, b, data_tar, pct, 0.1, 0.8, verbose)
    merged = merge_graphs(data_src, data_tar, ec, e_s, e_t)
    if mapper_idx_src is not None:
        merged = reorder(merged, data_src, mapper_idx_src, mapper_idx_tar)
    if save_path is not None:
        from .data import save_bridged_graph
        save_bridged_graph(merged, save_path)
    return merged
