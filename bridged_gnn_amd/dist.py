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
      3. halo exchange of the TRANSFORMED rows actually referenced by local in-edges: ONE
         `all_to_all_single` with uneven splits per conv carrying the rows of both feature tables,
         received straight into the tail of the per-conv allocation [h_s2t local | h_t2s local | halo]
         (so each table sees the halo as its own continuation).  Full mesh: every peer pair uses its
         own xGMI link, no ring;
      4. fused aggregation in two parts: every row's LOCAL-source edges are aggregated WHILE the halo is in
         flight in ONE launch (interior rows are finished there, rows with remote in-neighbours park their
         online-softmax state: (max, sum) + raw accumulator), the remote-source edges of those rows after it landed
         (`part` = 1 / 2 and `park_begin` of the aggregation ABI);
      5. outputs stay partitioned; BN(eval)/ReLU/log_softmax are row-local.
  * `PartitionPlan` (pure numpy + torch index tensors, device agnostic) is the host logic and is what the
    world_size-2 gloo CPU tests exercise; `PartitionedKTGNN` is the GPU driver on top of ops.py.
"""
import weakref

import numpy as np
import torch
import torch.distributed as dist

__all__ = ["partition_nodes", "PartitionPlan", "HaloExchange", "PartitionedKTGNN", "all_gather_rows", "shard_range"]


def shard_range(n, rank, world):
    """contiguous shard [lo, hi) of n rows for `rank` of `world` (the kNN bridge's query / candidate split)"""
    return rank * n // world, (rank + 1) * n // world


def all_gather_rows(t, group=None, world=None, always=False):
    """Row blocks of (possibly) different heights, one per rank -> their concatenation in rank order on every rank:
    one tiny all_gather of the heights + ONE all_gather of the padded payload (RCCL: direct full mesh over xGMI, every
    peer pair its own link).  The kNN bridge's only collective (`all_gather` of q_cand).  A gloo group
    with CUDA tensors (several ranks rehearsing on one GPU) stages the payload through the host."""
    if not dist.is_initialized():
        if (world is not None and world > 1) or always:
            # a shard of the candidates would otherwise be scored as if it were the whole set -- silently wrong edges
            raise RuntimeError(f"all_gather_rows(world={world}, always={always}) needs an initialised torch.distributed "
                               "process group")
        return t
    if world is not None and world != dist.get_world_size(group):
        raise RuntimeError(f"all_gather_rows: world={world} but the process group has {dist.get_world_size(group)} ranks")
    if dist.get_world_size(group) == 1 and not always:
        return t            # `always`: issue the collectives even at world size 1 (smoke-tests the RCCL calls)
    world = dist.get_world_size(group)
    host = t.is_cuda and dist.get_backend(group) == "gloo"
    work = t.cpu() if host else t.contiguous()
    h = torch.tensor([work.shape[0]], dtype=torch.int64, device=work.device)
    hs = [torch.empty_like(h) for _ in range(world)]
    dist.all_gather(hs, h, group=group)
    heights = [int(v.item()) for v in hs]
    hmax = max(heights)
    pad = work
    if work.shape[0] < hmax:
        pad = torch.zeros((hmax,) + tuple(work.shape[1:]), dtype=work.dtype, device=work.device)
        pad[: work.shape[0]] = work
    out = torch.empty((world * hmax,) + tuple(work.shape[1:]), dtype=work.dtype, device=work.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    if any(v != hmax for v in heights):
        out = torch.cat([out[r * hmax: r * hmax + heights[r]] for r in range(world)], dim=0)
    return out.to(t.device) if host else out


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

        # ---- ONE combined exchange per conv.  Per-rank memory of a conv's two tables:
        #        [ h_s2t local rows | h_t2s local rows | halo rows (peer-major; per peer: h_t2s rows, then h_s2t rows) ]
        #      so both tables see the halo as a continuation of themselves: from the h_t2s base (row n_local) a halo
        #      row sits at n_local + pos, from the h_s2t base (row 0) at 2*n_local + pos.  Halo/send lists are sorted
        #      by (peer, table, global id) on both sides, identically on every rank.
        sel = k_needer == rank                                   # rows I receive: sorted (table, owner, id) -> reorder
        r_id, r_owner, r_table = k_id[sel], k_owner[sel], k_table[sel]
        order = np.lexsort((r_id, r_table, r_owner))             # (owner, table, id)
        r_id, r_owner, r_table = r_id[order], r_owner[order], r_table[order]
        self.recv_splits = np.bincount(r_owner, minlength=world).astype(np.int64).tolist()
        self.n_halo = int(r_id.shape[0])
        self.halo_global = r_id.astype(np.int64)                 # global node id of every halo slot (a node needed from
        self.halo_mask = mask[r_id]                              # both tables holds two slots) and its domain
        self.n_halo_by_table = [int((r_table == 0).sum()), int((r_table == 1).sum())]
        halo_pos = [np.full(N, -1, dtype=np.int64), np.full(N, -1, dtype=np.int64)]
        pos = np.arange(r_id.shape[0], dtype=np.int64)
        for t in (0, 1):
            halo_pos[t][r_id[r_table == t]] = pos[r_table == t]
        sel = k_owner == rank                                    # rows I send: order (needer, table, id)
        s_id, s_needer, s_table = k_id[sel], k_needer[sel], k_table[sel]
        order = np.lexsort((s_id, s_table, s_needer))
        s_id, s_needer, s_table = s_id[order], s_needer[order], s_table[order]
        self.send_splits = np.bincount(s_needer, minlength=world).astype(np.int64).tolist()
        # index into the [h_s2t local | h_t2s local] region: h_t2s (table 0) rows live at n_local + r
        self.send_rows = g2l[s_id] + np.where(s_table == 0, self.n_local, 0)
        self.send_rows_local = g2l[s_id]                         # the same rows as plain local row numbers (input features)
        # ---- all-gather fallback ("a plain all_gather of [N/P, D] blocks when the halo ~ N").  Every rank
        #      contributes its two tables in ascending-global-id order (h_s2t block, then h_t2s block, padded to the largest
        #      rank); a halo slot (owner o, table t, node v) is then row  o * 2 * nmax + (t == 0) * nmax + rank_in_owner[v]
        counts = np.bincount(owner, minlength=world).astype(np.int64)
        self.n_local_max = int(counts.max())
        order_all = np.argsort(owner, kind="stable")             # nodes grouped by owner, ascending id inside a group
        rank_in_owner = np.empty(N, dtype=np.int64)
        rank_in_owner[order_all] = np.arange(N) - np.repeat(np.cumsum(counts) - counts, counts)
        self.gather_index = (r_owner.astype(np.int64) * 2 * self.n_local_max + np.where(r_table == 0, self.n_local_max, 0)
                             + rank_in_owner[r_id])
        self.sorted_local = g2l[mine]                            # local row of the k-th owned node in ascending-id order
        # payload of the two forms in rows: pack + all_to_all moves n_halo rows in, the all-gather 2 * nmax * (world - 1)
        self.halo_fraction = self.n_halo / max(2.0 * self.n_local_max * max(world - 1, 1), 1.0)
        # ---- local CSR (stable: input order inside a row, self loop last) ---------------------------
        keep = o_dst == rank
        ls, ld, lt = src[keep], dst[keep], table[keep]
        lrow = g2l[ld]
        halo_col = np.where(lt == 0, self.n_local + halo_pos[0][ls], 2 * self.n_local + halo_pos[1][ls])
        lcol = np.where(owner[ls] == rank, g2l[ls], halo_col)
        assert (lcol >= 0).all()
        order = np.argsort(lrow, kind="stable")
        self.col = lcol[order].astype(np.int32)
        # "extended" numbering for a conv whose halo INPUT rows are resident (two separate tables of n_local + n_halo
        # rows each, halo slot p at row n_local + p of the table that needs it)
        # ... with the slots regrouped by table (h_t2s-only rows, then h_s2t-only rows) so that the transform can give
        # each group just the table it is read from (`tail_single`)
        self.halo_ext_perm = np.concatenate([np.nonzero(r_table == 0)[0], np.nonzero(r_table == 1)[0]])   # ext order -> slot
        ext_pos = np.empty(max(self.n_halo, 1), dtype=np.int64)
        ext_pos[self.halo_ext_perm] = np.arange(self.n_halo)
        slot = np.where(lt == 0, halo_pos[0][ls], halo_pos[1][ls])
        lcol_ext = np.where(owner[ls] == rank, g2l[ls], self.n_local + ext_pos[np.maximum(slot, 0)])
        self.col_ext = lcol_ext[order].astype(np.int32)
        rp = np.zeros(self.n_local + 1, dtype=np.int64)
        np.add.at(rp, lrow + 1, 1)
        self.rowptr = np.cumsum(rp).astype(np.int32)
        self.local_num_edges = int(self.col.shape[0])
        # ---- the same rows split by where the SOURCE lives: local-source edges can be aggregated while the halo
        #      is still in flight (online-softmax state parked per row), remote-source edges after it landed
        srow, scol = lrow[order], lcol[order]
        is_remote = scol >= self.n_local
        self.rowptr_L, self.col_L = self._csr(srow[~is_remote], scol[~is_remote])
        self.rowptr_R, self.col_R = self._csr(srow[is_remote], scol[is_remote])

    def _csr(self, rows, cols):
        rp = np.zeros(self.n_local + 1, dtype=np.int64)
        np.add.at(rp, rows + 1, 1)
        return np.cumsum(rp).astype(np.int32), cols.astype(np.int32)      # rows are already sorted (stable)

    def table_views(self, big):
        """big: [2*n_local + n_halo, ld] -> (h_t2s view, h_s2t view) whose row indices match `col`."""
        return big[self.n_local:], big

    def summary(self):
        return {"rank": self.rank, "n_local": self.n_local, "n_interior": self.n_interior,
                "n_halo": self.n_halo, "n_halo_by_table": self.n_halo_by_table, "local_edges": self.local_num_edges}


class HaloExchange:
    """Device-agnostic exchange of transformed rows (nccl/RCCL on GPU tensors, gloo on CPU tensors): ONE
    `all_to_all_single` with uneven splits per conv, posted asynchronously by `start`, awaited by `wait`;
    received rows land directly in big[2*n_local:]."""

    ALLGATHER_FROM = 0.6      # `mode="auto"`: all-gather once the halo holds >= 60 % of everybody else's rows (both tables)

    def __init__(self, plan, device, group=None, always=False, mode="auto"):
        """mode: "a2a" = row pack + all_to_all of the referenced rows; "allgather" = every rank's whole [2 n_local, ld] block
        (SURVEY 8(e)'s fallback for graphs whose halo is ~ everything: no pack kernel, no uneven splits, one dense collective, at
        up to 1 / halo_fraction times the bytes); "auto" picks by `plan.halo_fraction`."""
        self.plan, self.group, self.always = plan, group, always   # always: issue the collectives even at world 1
        self.mode = mode if mode != "auto" else ("allgather" if plan.halo_fraction >= self.ALLGATHER_FROM and plan.world > 1 else "a2a")
        self.send_rows = torch.from_numpy(plan.send_rows).to(device)
        self.gather_index = torch.from_numpy(plan.gather_index).to(device)
        self.sorted_rows = torch.cat((torch.from_numpy(plan.sorted_local), torch.from_numpy(plan.sorted_local) + plan.n_local)).to(device)
        self._work, self._keep = None, None
        # gloo cannot move device memory: when the process group is gloo but the tables live on a GPU (debug /
        # single-GPU multi-process rehearsals) the payload is staged through the host.  RCCL never takes this path.
        self.host_staging = (torch.device(device).type == "cuda" and dist.is_initialized()
                             and dist.get_backend(group) == "gloo")

    def start(self, big):
        """big = [h_s2t local | h_t2s local | halo] ([2*n_local + n_halo, ld]); the local rows must be final."""
        p = self.plan
        if self.mode == "allgather" and (p.world > 1 or self.always):
            return self._start_allgather(big)
        if big.is_cuda and big.dtype == torch.float32 and big.shape[1] % 4 == 0 and big.stride(1) == 1:
            from . import ops
            send = ops.gather_rows(big, self.send_rows)              # [sum(send_splits), ld]
        else:                                                        # host tensors (gloo tests of the host logic)
            send = big.index_select(0, self.send_rows)
        recv = big[2 * p.n_local: 2 * p.n_local + p.n_halo]
        self._keep = send
        self._work = None
        if p.world == 1 and not self.always:
            return
        if self.host_staging:
            r_host = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(r_host, send.cpu(), output_split_sizes=p.recv_splits,
                                   input_split_sizes=p.send_splits, group=self.group)
            recv.copy_(r_host)
            return
        self._work = dist.all_to_all_single(recv, send, output_split_sizes=p.recv_splits,
                                            input_split_sizes=p.send_splits, group=self.group, async_op=True)

    def _start_allgather(self, big):
        """the fallback: [h_s2t sorted | h_t2s sorted] of every rank in one all_gather; the referenced rows are picked out of
        the gathered buffer in `wait` (the aggregation of the local-source edges runs in between, as with the all_to_all)"""
        p = self.plan
        ld = big.shape[1]
        blk = torch.zeros(2 * p.n_local_max, ld, dtype=big.dtype, device=big.device)
        rows = big.index_select(0, self.sorted_rows)                 # [2 n_local, ld]: h_s2t rows, then h_t2s rows, ascending ids
        blk[:p.n_local] = rows[:p.n_local]
        blk[p.n_local_max:p.n_local_max + p.n_local] = rows[p.n_local:]
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if self.host_staging:
            out = torch.empty(world * 2 * p.n_local_max, ld, dtype=big.dtype)
            dist.all_gather_into_tensor(out, blk.cpu(), group=self.group)
            self._gathered = (out.to(big.device), big)
            self._work = None
        else:
            out = torch.empty(world * 2 * p.n_local_max, ld, dtype=big.dtype, device=big.device)
            self._work = dist.all_gather_into_tensor(out, blk, group=self.group, async_op=True)
            self._gathered = (out, big)
        self._keep = blk

    def wait(self):
        if self._work is not None:
            self._work.wait()
        g = getattr(self, "_gathered", None)
        if g is not None:
            out, big = g
            p = self.plan
            big[2 * p.n_local: 2 * p.n_local + p.n_halo] = out.index_select(0, self.gather_index)
            self._gathered = None
        self._work, self._keep = None, None

    def exchange_rows(self, rows):
        """rows [sum(send_splits), ld] (already in send order) -> [n_halo, ld] in halo order; blocking.  Used once per
        version of the input features (`PartitionedKTGNN._input_ext`), not per forward."""
        p = self.plan
        recv = torch.empty(p.n_halo, rows.shape[1], dtype=rows.dtype, device=rows.device)
        if p.world == 1 and not self.always:
            return recv
        if self.host_staging:
            r_host = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(r_host, rows.cpu(), output_split_sizes=p.recv_splits,
                                   input_split_sizes=p.send_splits, group=self.group)
            recv.copy_(r_host)
            return recv
        dist.all_to_all_single(recv, rows.contiguous(), output_split_sizes=p.recv_splits,
                               input_split_sizes=p.send_splits, group=self.group)
        return recv


class PartitionedKTGNN:
    """Eval forward of `KTGNN_no_complement` (models/KTGNN.py:401-435) on rank-local rows."""

    def __init__(self, model, edge_index, central_mask, rank, world, device, owner=None, group=None,
                 always_communicate=False, cache_input_halo=True, halo_mode="auto"):
        from . import ops
        # cache_input_halo: the FIRST conv reads the graph's input features, which do not change between forwards (the
        # reference trains 300 epochs on one `data.x`).  Their halo rows are fetched once per version of x (same send
        # lists, same all_to_all) and kept next to the local rows; every forward then transforms local + halo rows itself
        # (weights do change) and needs no per-forward exchange of 512-byte rows for that conv -- on C4 / 8 ranks that
        # exchange is ~100 MB per rank per forward, more than the rank's whole compute.  Later convs consume activations
        # and keep the per-forward exchange.  False = exchange transformed rows for every conv.
        self.cache_input_halo = bool(cache_input_halo)
        # a resident halo row feeds destinations of ONE domain: the transform computes just that table for it (one launch,
        # the other table's waves sit those row tiles out: GemmParams::tail_*; hidden transform of a rank's share 132 -> 9x us)
        self.single_table_halo = True
        self._x_ext_key, self._x_ext = None, None
        self._x_sums_key, self._x_sums = None, None
        self.model, self.rank, self.world, self.device, self.group = model, rank, world, device, group
        self.always = always_communicate               # run the collectives even at world_size 1 (smoke-tests RCCL usage)
        plan = PartitionPlan(edge_index, central_mask, rank, world, owner=owner)
        self.plan = plan
        self.owned_global = torch.from_numpy(plan.owned_global).to(device)
        self.global_num_edges, self.local_num_edges = plan.global_num_edges, plan.local_num_edges
        t = lambda a: torch.from_numpy(a).to(device)
        self.csr_L = ops.DstCSR(t(plan.rowptr_L), t(plan.col_L), None, int(plan.col_L.shape[0]), plan.n_local)
        self.csr_R = ops.DstCSR(t(plan.rowptr_R), t(plan.col_R), None, int(plan.col_R.shape[0]), plan.n_local)
        self.csr_ext = ops.DstCSR(t(plan.rowptr), t(plan.col_ext), None, plan.local_num_edges, plan.n_local)
        self.send_rows_local = t(plan.send_rows_local)
        self.halo_ext_perm = t(plan.halo_ext_perm)
        self.mask_ext_u8 = torch.cat((t(plan.mask_local), t(plan.halo_mask[plan.halo_ext_perm]))).to(torch.uint8).contiguous()
        self._states = {}
        self._state3 = None
        self.mask_local = torch.from_numpy(plan.mask_local).to(device)
        self.mask_u8 = self.mask_local.to(torch.uint8).contiguous()
        self.halo = HaloExchange(plan, device, group, always=always_communicate, mode=halo_mode)

    def _all_reduce(self, t):
        if self.halo.host_staging:                                   # gloo rehearsal on GPU tensors (see HaloExchange)
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            return h.to(t.device)
        dist.all_reduce(t, group=self.group)
        return t

    def _aggregate_two_part(self, tables_list, convs, outs, ep=(None, None, False), colsum=None):
        """(1) local-source edges of every row (interior rows finish, boundary rows park their state) while the
        exchange is in flight; (2) wait; (3) remote-source edges of the boundary rows.  Lists = several convs that
        share the exchange."""
        from . import ops
        p = self.plan
        single = not isinstance(convs, (list, tuple))
        if single:
            tables_list, convs, outs = [tables_list], [convs], [outs]
        sc, sh, relu = ep
        args = []
        for (h_t2s, h_s2t), conv in zip(tables_list, convs):
            args.append((h_t2s, h_s2t, conv.a_f_t2s.weight.detach().reshape(-1).contiguous(),
                         conv.a_f_s2t.weight.detach().reshape(-1).contiguous(), conv.out_channels, conv.negative_slope))
        kw = dict(n_dst=p.n_local, ep_scale=sc, ep_shift=sh, ep_relu=relu)
        for (ht, hs, a1, a2, D, slope), out in zip(args, outs):
            # ONE launch over the local-source edges of every row: interior rows (< n_interior) are finished,
            # boundary rows park their state
            ops.adaptedconv_aggregate(ht, hs, a1, a2, self.csr_L, self.mask_u8, D, slope, out=out,
                                      row_begin=0, row_end=p.n_local, state_ms=self._state(out), part=1,
                                      park_begin=p.n_interior, colsum=colsum, **kw)
        self.halo.wait()
        for (ht, hs, a1, a2, D, slope), out in zip(args, outs):
            ops.adaptedconv_aggregate(ht, hs, a1, a2, self.csr_R, self.mask_u8, D, slope, out=out,
                                      row_begin=p.n_interior, row_end=p.n_local, state_ms=self._state(out), part=2,
                                      colsum=colsum, **kw)

    def _state(self, out):
        """(max, sum) scratch of the rows parked between the two parts, one per output buffer in flight."""
        key = out.data_ptr()
        st = self._states.get(key)
        if st is None or st.shape[0] < self.plan.n_local:
            st = torch.empty(self.plan.n_local, 2, dtype=torch.float32, device=self.device)
            self._states = {**self._states, key: st} if len(self._states) < 8 else {key: st}
        return st

    def _input_ext(self, x):
        """[x local rows ; x halo rows] (columns padded to a multiple of 4) for the current version of the input features:
        the halo is fetched on first use and again whenever x is another tensor object or was written in place."""
        from .ktgnn import _pad_cols4
        key = self._x_ext_key
        if key is None or key[0]() is not x or key[1] != x._version:
            xp = _pad_cols4(x)
            halo = self.halo.exchange_rows(xp.index_select(0, self.send_rows_local))
            self._x_ext = torch.cat((xp, halo.index_select(0, self.halo_ext_perm)))      # halo regrouped by table
            self._x_ext_key = (weakref.ref(x), x._version)
        return self._x_ext

    def invalidate_input_cache(self):
        """forget the cached input halo rows and the all-reduced domain sums of `x` (both keyed by tensor identity +
        `_version`): call it after a write to x that does not advance the version (`x.data[...] = ...`, an alias, a
        raw-pointer kernel) -- same contract as `KTGNN_no_complement.invalidate_input_cache`."""
        self._x_sums_key, self._x_sums = None, None
        self._x_ext_key = None

    def _conv_resident_halo(self, conv, x, epilogue=None, out_sums=None, arena=None):
        """First conv with resident input halo: all-reduce of the domain sums, transform of local + halo rows, ONE
        aggregation launch over the complete local CSR -- no per-forward row exchange."""
        from . import ops
        from .ktgnn import _pad_cols4
        p = self.plan
        xp = _pad_cols4(x)
        # the (all-reduced) domain sums of the static input features are kept with their halo: per version of x one
        # stream over the local rows and ONE all-reduce, not one of each per forward
        skey = self._x_sums_key
        if skey is None or skey[0]() is not x or skey[1] != x._version:
            sums = ops.domain_sums(xp, self.mask_u8)
            if self.world > 1 or self.always:
                sums = self._all_reduce(sums)
            self._x_sums, self._x_sums_key = sums, (weakref.ref(x), x._version)
        sums = self._x_sums
        # which table(s) a 32-row tile of [own rows ; halo rows] is ever read from (a halo row feeds destinations of one domain;
        # with s -> t bridge edges a target row's h_t2s is dead as well): generalises tail_single
        need = self.csr_ext.tile_need(self.mask_u8, table_mask_u8=self.mask_ext_u8) if (self.single_table_halo and conv.out_channels > 32) else None
        if need is not None:
            h_t2s, h_s2t = conv.transform(self._input_ext(x), self.mask_ext_u8, sums=sums, tile_need=need)
        else:
            h_t2s, h_s2t = conv.transform(self._input_ext(x), self.mask_ext_u8, sums=sums,
                                          tail_single=tuple(p.n_halo_by_table) if self.single_table_halo else (0, 0))
        sc, sh, relu = epilogue if epilogue is not None else (None, None, False)
        out = ops.adaptedconv_aggregate(h_t2s, h_s2t, conv.a_f_t2s.weight.detach().reshape(-1).contiguous(),
                                        conv.a_f_s2t.weight.detach().reshape(-1).contiguous(), self.csr_ext, self.mask_u8,
                                        conv.out_channels, conv.negative_slope, n_dst=p.n_local,
                                        ep_scale=sc, ep_shift=sh, ep_relu=relu, colsum=out_sums)
        return out[:, : conv.out_channels], sums

    def _conv(self, conv, x, epilogue=None, sums=None, out_sums=None, arena=None):
        from . import ops
        from .ktgnn import _pad_cols4
        p = self.plan
        xp = _pad_cols4(x)
        if sums is None:
            sums = ops.domain_sums(xp, self.mask_u8, out=arena.take(2 * xp.shape[1] + 2) if arena is not None else None)
            if self.world > 1 or self.always:
                sums = self._all_reduce(sums)                        # 2*Din+2 doubles
        ld = ops.pad4(conv.out_channels)
        # one allocation per conv: [h_s2t local | h_t2s local | halo]; the transform writes both local parts
        big = torch.empty(2 * p.n_local + p.n_halo, ld, dtype=torch.float32, device=self.device)
        h_t2s, h_s2t = p.table_views(big)
        conv.transform(x, self.mask_u8, sums=sums, out=(h_t2s, h_s2t))     # writes rows [0, n_local) of each view
        self.halo.start(big)
        a_t2s = conv.a_f_t2s.weight.detach().reshape(-1).contiguous()
        a_s2t = conv.a_f_s2t.weight.detach().reshape(-1).contiguous()
        sc, sh, relu = epilogue if epilogue is not None else (None, None, False)
        out = torch.empty(p.n_local, ops.pad4(conv.out_channels), dtype=torch.float32, device=self.device)
        self._aggregate_two_part((h_t2s, h_s2t), conv, out, ep=(sc, sh, relu), colsum=out_sums)
        return out[:, : conv.out_channels], sums

    @torch.no_grad()
    def forward(self, x_local):
        """x_local = x[owned_global] (rank-local rows in plan order) -> (logp_base, logp_target,
        logp_target_hat) for those rows.  With `cache_input_halo` the halo rows of x are re-fetched whenever x_local is
        a different tensor or was modified in place (`_version`); a captured HIP graph cannot see that check, so
        re-capture after changing x."""
        import torch.nn.functional as F
        m = self.model
        if m.training:
            raise NotImplementedError("partitioned forward is eval-only (BN batch statistics would need an all-reduce)")
        from . import ops
        x = x_local.float().contiguous()
        s_h = both = None
        # every float64 accumulator of this forward comes out of ONE zero fill; the two sums that travel in the
        # classifier stage's all-reduce (of h and of T's hidden activation) sit next to each other in it
        width = max([x.shape[1]] + [c.out_channels for c in m.convs])
        arena = ops.ZeroArena(self.device, (len(m.convs) + 3) * (2 * ops.pad4(width) + 2))
        for ind, conv in enumerate(m.convs):
            if m.use_bn:
                from .ktgnn import bn_eval_affine
                sc, sh = bn_eval_affine(m.bns[ind])
                last = ind == len(m.convs) - 1
                # fused sums in the epilogue (one pass fewer over the rank-local activations; neutral on one GPU)
                if last:
                    n_h = 2 * ops.pad4(conv.out_channels) + 2
                    both = arena.take(2 * n_h)
                    s_h = both[:n_h]
                run = self._conv_resident_halo if (ind == 0 and self.cache_input_halo) else self._conv
                x, _ = run(conv, x, epilogue=(sc, sh, True), out_sums=s_h, arena=arena)   # epilogue also sums the finished rows
            else:
                run = self._conv_resident_halo if (ind == 0 and self.cache_input_halo) else self._conv
                x, _ = run(conv, x, arena=arena)
                x = F.relu(x)
            x = x.contiguous()
        # the two classifier inputs (h and T(h)) are both row-local once the hidden conv is done: their domain
        # sums travel in ONE all-reduce
        from . import ops
        from .ktgnn import _pad_cols4
        # h1; T's last Linear is folded into the conv weights; its rank-local domain sums come out of the GEMM epilogue
        adjacent = both is not None and x.shape[1] == m.clf_transformer[0].weight.shape[0]
        m._fold_transformer()
        dout, din = m._tf_w0.shape
        pack_t = m._composed_target_pack(ops.pad4(dout))
        raw = None
        if not (self.world > 1 or self.always) and s_h is not None:
            # nothing to all-reduce: the three convs' narrow tables in ONE pass over h (bgnn_classifier_stage_f32), as the single-GPU
            # forward does.  (With ranks the fused kernel would need the all-reduced sums of h BEFORE the pass and of T's hidden
            # activation AFTER it -- two all-reduces instead of one; a rank's second pass over its 1/N of h costs less.)
            res = self._classifier_stage(x, None, s_h, None, arena=arena, fuse=True)
            if res is not None:
                logp, fused = res
                if not fused:
                    logp = F.log_softmax(logp, dim=2)
                return logp[:, 0], logp[:, 1], logp[:, 2]
        if (adjacent and x.dtype == torch.float32 and x.stride(1) == 1 and x.shape[1] == din
                and ops.linear_narrow_supported(din, dout, pack_t)):
            # fused pair: h1 never reaches HBM; stage A leaves 12 floats per row + the rank-local sums of h1, stage B
            # (in _classifier_stage, after the all-reduce) finishes the narrow tables
            s_t = both[s_h.numel():]
            raw = ops.linear_narrow_transform(x, m._tf_w0, m._tf_b0, self.mask_u8, s_t, pack_t, relu=True)
            xt = None
        else:
            xt, s_t = m._transformer_hidden_eval(x, self.mask_u8, want_sums=True, sums_out=both[s_h.numel():] if adjacent else None)
            xt = xt.contiguous()
        if s_h is None:
            s_h = ops.domain_sums(_pad_cols4(x), self.mask_u8)
        if raw is None and (s_t is None or s_t.numel() != 2 * _pad_cols4(xt).shape[1] + 2):
            s_t = ops.domain_sums(_pad_cols4(xt), self.mask_u8)
        if not (adjacent and s_t.data_ptr() == both[s_h.numel():].data_ptr()):
            both = torch.cat((s_h, s_t))
        if self.world > 1 or self.always:
            both = self._all_reduce(both)
        s_h, s_t = both[: s_h.numel()], both[s_h.numel():]
        logp, fused = self._classifier_stage(x, xt, s_h, s_t, raw=raw, pack_t=pack_t)
        if not fused:
            logp = F.log_softmax(logp, dim=2)                                    # one launch for the three heads
        return logp[:, 0], logp[:, 1], logp[:, 2]

    def _classifier_stage(self, x, xt, s_h, s_t, raw=None, pack_t=None, arena=None, fuse=False):
        """clf_base(x), clf_target(x), clf_target(T(x)) (KTGNN.py:432-434; `xt` = hidden activation h1 of T, whose
        last Linear is folded into the packed weights) with ONE halo exchange: the six narrow
        tables are interleaved column-wise in one allocation (row = [base | target | target-hat] x pad4(C) floats),
        so a halo row carries all three convs' values."""
        from . import ops
        from .ktgnn import _pad_cols4
        m, p = self.model, self.plan
        C = m.clf_base.out_channels
        ld = ops.pad4(C)
        big = torch.empty(2 * p.n_local + p.n_halo, 3 * ld, dtype=torch.float32, device=self.device)
        views = [(big[p.n_local:, j * ld:(j + 1) * ld], big[:, j * ld:(j + 1) * ld]) for j in range(3)]   # (h_t2s, h_s2t)
        if fuse:
            if not m._classifier_stage_fused(x, self.mask_u8, s_h, views, arena):
                return None                                          # outside the kernel's envelope: the caller takes the separate launches
        else:
            m.clf_base.transform(x, self.mask_u8, sums=s_h, partner=m.clf_target, out=[views[0], views[1]])
            if raw is not None:                                      # stage B of the fused linear -> narrow transform
                ops.narrow_transform_finish(raw, self.mask_u8, s_t, pack_t, views[2])
            else:
                xtp = _pad_cols4(xt)
                ops.adaptedconv_transform(xtp, self.mask_u8, None, m._composed_target_pack(xtp.shape[1]), out=[views[2]], sums=s_t)
        self.halo.start(big)
        akey = (m.clf_base._versions(), m.clf_target._versions())
        if getattr(m, "_a3_key", None) != akey:                      # same cache as the single-GPU forward
            convs = (m.clf_base, m.clf_target, m.clf_target)
            m._a3 = (torch.stack([c.a_f_t2s.weight.detach().reshape(-1) for c in convs]).contiguous(),
                     torch.stack([c.a_f_s2t.weight.detach().reshape(-1) for c in convs]).contiguous())
            m._a3_key = akey
        a_t2s, a_s2t = m._a3
        out3 = torch.empty(p.n_local, 3 * ld, dtype=torch.float32, device=self.device)
        h_t2s, h_s2t = p.table_views(big)                            # interleaved [rows, 3*ld] tables
        st = self._state3
        if st is None or st.shape[0] < 3 * p.n_local:
            st = self._state3 = torch.empty(3 * p.n_local, 2, dtype=torch.float32, device=self.device)
        fused = ops.heads_log_softmax_supported(3, C)                # KTGNN.py:435 in the epilogue of the finishing launches
        kw = dict(n_dst=p.n_local, out=out3, heads=3, log_softmax=fused)
        slope = m.clf_base.negative_slope
        ops.adaptedconv_aggregate(h_t2s, h_s2t, a_t2s, a_s2t, self.csr_L, self.mask_u8, C, slope,
                                  row_begin=0, row_end=p.n_local, state_ms=st, part=1, park_begin=p.n_interior, **kw)
        self.halo.wait()
        ops.adaptedconv_aggregate(h_t2s, h_s2t, a_t2s, a_s2t, self.csr_R, self.mask_u8, C, slope,
                                  row_begin=p.n_interior, row_end=p.n_local, state_ms=st, part=2, **kw)
        return out3.view(p.n_local, 3, ld)[:, :, :C], fused          # [n_local, 3, C]: base, target, target-hat
