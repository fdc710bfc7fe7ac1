"""Multi-GPU KT-GNN: destination-node partitioning + halo exchange over RCCL (xGMI).

The reference is single-process / single-device; this is new design
(SURVEY.md 8(e)).  One process per GPU (`torch.distributed`, backend "nccl" == RCCL on ROCm).

  * Rank r owns a set of DESTINATION nodes and all their in-edges (its CSR row block).  Default
    partition = "domain blocks": the r-th contiguous block of the source domain plus the r-th block of
    the target domain, so the kNN bridge edges (target i <- sources near the same relative position)
    stay mostly rank-local and the halo is only the non-local fraction of the graph.
  * Per AdaptedConv (reference models/KTGNN.py:263-315):
      1. per-domain column sums of the local x -> ONE all-reduce of 2*Din+2 doubles -> delta (:275);
      2. local dense transform -> h_t2s / h_s2t rows of the owned nodes (:277-284);
      3. halo exchange of the TRANSFORMED rows actually referenced by local in-edges:
         `all_to_all_single` with uneven splits, one per feature table, received straight into the
         tail of the local table (rows [n_local, n_local + n_halo)).  Full mesh: every peer pair uses
         its own xGMI link, no ring;
      4. fused aggregation: interior rows (all in-neighbours local) are aggregated WHILE the halo is
         in flight, boundary rows after it lands (row-range launches of the same kernel);
      5. outputs stay partitioned; BN(eval)/ReLU/log_softmax are row-local.
  * `PartitionPlan` (pure numpy + torch index tensors, device agnostic) is the host logic and is what the
    world_size-2 gloo CPU tests exercise; `PartitionedKTGNN` is the GPU driver on top of ops.py.
"""
import numpy as np
import torch
import torch.distributed as dist

__all__ = ["partition_nodes", "PartitionPlan", "HaloExchange", "PartitionedKTGNN"]


def partition_nodes(central_mask, world, scheme="domain_blocks"):
    """-> int32 [N] owner rank of every node."""
    m = np.asarray(central_mask, dtype=bool)
    n = m.shape[0]
    owner = np.empty(n, dtype=np.int32)
    if scheme == "domain_blocks":
        for dom in (True, False):
            ids = np.nonzero(m == dom)[0]
            pos = np.arange(ids.shape[0], dtype=np.int64)
            owner[ids] = np.minimum((pos * world) // max(ids.shape[0], 1), world - 1).astype(np.int32)
    elif scheme == "contiguous":
        owner[:] = np.minimum((np.arange(n, dtype=np.int64) * world) // max(n, 1), world - 1)
    else:
        raise ValueError(scheme)
    return owner


class PartitionPlan:
    """Everything rank `rank` needs to run its row block: local CSR (interior rows first), halo
    numbering, all_to_all split sizes and send lists.  Built identically (deterministically) on every
    rank from the replicated edge list; no communication."""

    def __init__(self, edge_index, central_mask, rank, world, owner=None, rewrite_self_loops=True):
        ei = np.asarray(edge_index, dtype=np.int64)
        mask = np.asarray(central_mask, dtype=bool)
        N = mask.shape[0]
        self.rank, self.world, self.N = rank, world, N
        owner = partition_nodes(mask, world) if owner is None else np.asarray(owner, dtype=np.int32)
        self.owner = owner
        if rewrite_self_loops:                       # graph_partition, models/KTGNN.py:385-398
            ei = ei[:, ei[0] != ei[1]]
            loops = np.arange(N, dtype=np.int64)
            ei = np.concatenate([ei, np.stack([loops, loops])], axis=1)
        self.global_num_edges = int(ei.shape[1])
        src, dst = ei[0], ei[1]
        o_src, o_dst = owner[src], owner[dst]
        table = (~mask[dst]).astype(np.int64)        # 0: h_t2s (destination in S), 1: h_s2t (destination in T)

        # ---- who needs which remote row (all rank pairs; identical on every rank) -----------------
        remote = o_src != o_dst
        key = ((o_dst[remote].astype(np.int64) * 2 + table[remote]) * world + o_src[remote]) * N + src[remote]
        key = np.unique(key)                         # sorted: (needer, table, owner, global id)
        k_id = key % N
        k_rest = key // N
        k_owner = (k_rest % world).astype(np.int32)
        k_table = ((k_rest // world) % 2).astype(np.int32)
        k_needer = (k_rest // (2 * world)).astype(np.int32)

        mine = np.nonzero(owner == rank)[0]          # ascending global ids
        # ---- local row order: interior rows (no remote in-neighbour) first -------------------------
        has_remote = np.zeros(N, dtype=bool)
        np.logical_or.at(has_remote, dst[remote], True)
        interior = mine[~has_remote[mine]]
        boundary = mine[has_remote[mine]]
        self.owned_global = np.concatenate([interior, boundary])
        self.n_local, self.n_interior = int(mine.shape[0]), int(interior.shape[0])
        g2l = np.full(N, -1, dtype=np.int64)
        g2l[self.owned_global] = np.arange(self.n_local)
        self.mask_local = mask[self.owned_global]

        # ---- receive side: halo numbering per table ---------------------------------------------
        self.recv_splits, self.n_halo, halo_index = [], [], []
        for t in (0, 1):
            sel = (k_needer == rank) & (k_table == t)
            ids, owners = k_id[sel], k_owner[sel]     # already sorted by (owner, id)
            self.recv_splits.append(np.bincount(owners, minlength=world).astype(np.int64).tolist())
            self.n_halo.append(int(ids.shape[0]))
            hi = np.full(N, -1, dtype=np.int64)
            hi[ids] = self.n_local + np.arange(ids.shape[0])
            halo_index.append(hi)
        # ---- send side: rows of mine each peer needs, per table, ordered (peer, global id) ---------
        self.send_splits, self.send_rows = [], []
        for t in (0, 1):
            sel = (k_owner == rank) & (k_table == t)
            ids, needers = k_id[sel], k_needer[sel]
            order = np.lexsort((ids, needers))
            self.send_splits.append(np.bincount(needers, minlength=world).astype(np.int64).tolist())
            self.send_rows.append(g2l[ids[order]])
        # ---- local CSR (stable: input order inside a row, self loop last) ---------------------------
        keep = o_dst == rank
        ls, ld, lt = src[keep], dst[keep], table[keep]
        lrow = g2l[ld]
        lcol = np.where(owner[ls] == rank, g2l[ls], np.where(lt == 0, halo_index[0][ls], halo_index[1][ls]))
        assert (lcol >= 0).all()
        order = np.argsort(lrow, kind="stable")
        self.col = lcol[order].astype(np.int32)
        rp = np.zeros(self.n_local + 1, dtype=np.int64)
        np.add.at(rp, lrow + 1, 1)
        self.rowptr = np.cumsum(rp).astype(np.int32)
        self.local_num_edges = int(self.col.shape[0])

    def summary(self):
        return {"rank": self.rank, "n_local": self.n_local, "n_interior": self.n_interior,
                "n_halo": self.n_halo, "local_edges": self.local_num_edges}


class HaloExchange:
    """Device-agnostic exchange of transformed rows (works with nccl/RCCL on GPU tensors and with gloo on
    CPU tensors).  `start` posts both all_to_all_single ops asynchronously, `wait` blocks the current
    stream on them; received rows land directly in table[n_local:]."""

    def __init__(self, plan, device, group=None, always=False):
        self.plan, self.group, self.always = plan, group, always   # always: issue the collectives even at world 1
        self.send_rows = [torch.from_numpy(r).to(device) for r in plan.send_rows]
        self._work = []

    def start(self, tables):
        """tables = (h_t2s, h_s2t), each [n_local + n_halo[t], ld]; rows < n_local must be final."""
        p = self.plan
        self._work = []
        self._keep = []
        for t, tab in enumerate(tables):
            send = tab.index_select(0, self.send_rows[t])            # [sum(send_splits), ld]
            recv = tab[p.n_local: p.n_local + p.n_halo[t]]
            self._keep.append(send)
            if p.world == 1 and not self.always:
                continue
            w = dist.all_to_all_single(recv, send, output_split_sizes=p.recv_splits[t],
                                       input_split_sizes=p.send_splits[t], group=self.group, async_op=True)
            self._work.append(w)

    def wait(self):
        for w in self._work:
            w.wait()
        self._work, self._keep = [], []


class PartitionedKTGNN:
    """Eval forward of `KTGNN_no_complement` (models/KTGNN.py:401-435) on rank-local rows."""

    def __init__(self, model, edge_index, central_mask, rank, world, device, owner=None, group=None,
                 always_communicate=False):
        from . import ops
        self.model, self.rank, self.world, self.device, self.group = model, rank, world, device, group
        self.always = always_communicate               # run the collectives even at world_size 1 (smoke-tests RCCL usage)
        plan = PartitionPlan(edge_index, central_mask, rank, world, owner=owner)
        self.plan = plan
        self.owned_global = torch.from_numpy(plan.owned_global).to(device)
        self.global_num_edges, self.local_num_edges = plan.global_num_edges, plan.local_num_edges
        self.csr = ops.DstCSR(torch.from_numpy(plan.rowptr).to(device), torch.from_numpy(plan.col).to(device), None,
                              plan.local_num_edges, plan.n_local)
        self.mask_local = torch.from_numpy(plan.mask_local).to(device)
        self.mask_u8 = self.mask_local.to(torch.uint8).contiguous()
        self.halo = HaloExchange(plan, device, group, always=always_communicate)

    def _conv(self, conv, x, epilogue=None, sums=None):
        from . import ops
        from .ktgnn import _pad_cols4
        p = self.plan
        xp = _pad_cols4(x)
        if sums is None:
            sums = ops.domain_sums(xp, self.mask_u8)
            if self.world > 1 or self.always:
                dist.all_reduce(sums, group=self.group)              # 2*Din+2 doubles
        delta = ops.domain_delta(sums, xp.shape[1])
        ld = ops.pad4(conv.out_channels)
        # local rows followed by halo rows in one allocation per table; the transform writes the local part
        h_t2s = torch.empty(p.n_local + p.n_halo[0], ld, dtype=torch.float32, device=self.device)
        h_s2t = torch.empty(p.n_local + p.n_halo[1], ld, dtype=torch.float32, device=self.device)
        conv.transform(x, self.mask_u8, delta=delta, out=(h_t2s, h_s2t))   # writes rows [0, n_local)
        self.halo.start((h_t2s, h_s2t))
        a_t2s = conv.a_f_t2s.weight.detach().reshape(-1).contiguous()
        a_s2t = conv.a_f_s2t.weight.detach().reshape(-1).contiguous()
        sc, sh, relu = epilogue if epilogue is not None else (None, None, False)
        out = torch.empty(p.n_local, ops.pad4(conv.out_channels), dtype=torch.float32, device=self.device)
        kw = dict(n_dst=p.n_local, ep_scale=sc, ep_shift=sh, ep_relu=relu, out=out)
        # interior rows overlap with the exchange; boundary rows need the halo
        ops.adaptedconv_aggregate(h_t2s, h_s2t, a_t2s, a_s2t, self.csr, self.mask_u8, conv.out_channels,
                                  conv.negative_slope, row_begin=0, row_end=p.n_interior, **kw)
        self.halo.wait()
        ops.adaptedconv_aggregate(h_t2s, h_s2t, a_t2s, a_s2t, self.csr, self.mask_u8, conv.out_channels,
                                  conv.negative_slope, row_begin=p.n_interior, row_end=p.n_local, **kw)
        return out[:, : conv.out_channels], sums

    @torch.no_grad()
    def forward(self, x_local):
        """x_local = x[owned_global] (rank-local rows in plan order) -> (logp_base, logp_target,
        logp_target_hat) for those rows."""
        import torch.nn.functional as F
        m = self.model
        if m.training:
            raise NotImplementedError("partitioned forward is eval-only (BN batch statistics would need an all-reduce)")
        x = x_local.float().contiguous()
        for ind, conv in enumerate(m.convs):
            if m.use_bn:
                bn = m.bns[ind]
                sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().float().contiguous()
                sh = (bn.bias - bn.running_mean * sc).detach().float().contiguous()
                x, _ = self._conv(conv, x, epilogue=(sc, sh, True))
            else:
                x, _ = self._conv(conv, x)
                x = F.relu(x)
            x = x.contiguous()
        base, sums = self._conv(m.clf_base, x)
        hat, _ = self._conv(m.clf_target, m._transformer_eval(x).contiguous())
        targ, _ = self._conv(m.clf_target, x, sums=sums)
        return F.log_softmax(base, dim=1), F.log_softmax(targ, dim=1), F.log_softmax(hat, dim=1)
