"""Tensor-level wrappers over the C ABI (one Python function per entry point of include/bgnn.h).
Every function takes/returns CUDA(HIP) tensors on the current device and launches on torch's
current stream.  No CPU path exists: host tensors raise."""
import os

import torch

from . import _lib as L

__all__ = ["DstCSR", "build_dst_csr", "domain_delta", "pack_transform_heads", "adaptedconv_transform", "adaptedconv_aggregate", "linear", "linear_supported", "linear_narrow_supported", "linear_narrow_transform", "narrow_transform_finish", "gram", "gram_supported", "rowdot", "transform_bwd_prep", "topk_edges_coalesced",
           "l2_normalize_rows", "cosine_topk", "mlp_pair_topk", "topk_edges", "coalesce", "gather_rows", "pad4"]


def pad4(n):
    return (int(n) + 3) // 4 * 4


# rows with at least this many in-edges are cut into segments of HUB_SEGMENT edges (AggParams::hub_threshold); tools/hub_sweep.py:
# config 3 (128, 64) 0.313 = (192, 64) 0.313, (192, 32) 0.330, (64, 32) 0.366 ms; config 2 (128, 64) 0.144, (192, 64) 0.155, off 0.162 ms
HUB_THRESHOLD = 128
HUB_SEGMENT = 64


class DstCSR:
    """By-destination CSR of the rewritten edge set: the build's replacement for the cached
    (edge_index1, edge_index2) pair of `KTGNN_no_complement.graph_partition`
    (reference models/KTGNN.py:385-398, :409-412).  Domain of a row = central_mask[row]."""

    def __init__(self, rowptr, col, eperm, num_edges, num_nodes):
        self.rowptr, self.col, self.eperm = rowptr, col, eperm
        self.num_edges, self.num_nodes = int(num_edges), int(num_nodes)
        self._transposed = None
        self._hubs = {}

    @staticmethod
    def _hub_tables_of(rowptr, N, threshold, segment):
        rp = rowptr[:N + 1].long()
        deg = rp[1:] - rp[:-1]
        hubs = torch.nonzero(deg >= threshold).reshape(-1)               # one-time sync, like the CSR build itself
        if hubs.numel() == 0:
            return None
        nseg = (deg[hubs] + segment - 1) // segment                       # segments per hub row
        seg_ptr = torch.zeros(hubs.numel() + 1, dtype=torch.int64, device=hubs.device)
        seg_ptr[1:] = torch.cumsum(nseg, 0)
        total = int(seg_ptr[-1].item())
        seg_hub = torch.repeat_interleave(torch.arange(hubs.numel(), device=hubs.device), nseg)   # hub index of a segment
        seg_in_hub = torch.arange(total, device=hubs.device) - seg_ptr[seg_hub]
        start = rp[hubs][seg_hub] + seg_in_hub * segment
        end = torch.minimum(start + segment, rp[hubs + 1][seg_hub])
        # hub rows are not adjacent in the edge arrays, so a segment carries its own (begin, end) pair: v -> bounds[2v], bounds[2v+1]
        bounds = torch.stack((start, end), dim=1).reshape(-1)
        return (hubs.to(torch.int32).contiguous(), seg_ptr.to(torch.int32).contiguous(),
                bounds.to(torch.int32).contiguous(), hubs[seg_hub].to(torch.int32).contiguous())

    def hub_tables(self, threshold=HUB_THRESHOLD, segment=HUB_SEGMENT):
        """Rows with >= `threshold` in-edges, cut into segments of <= `segment` edges, built once per graph:
        None when the graph has no such row, else (hub_rows [nh], hub_seg_ptr [nh+1], seg_bounds [2 nseg] = (begin, end) edge
        offsets into `col` per segment, seg_node [nseg]), int32 device tensors (see bgnn_adaptedconv_aggregate_hub_f32)."""
        key = (int(threshold), int(segment))
        if key not in self._hubs:
            self._hubs[key] = DstCSR._hub_tables_of(self.rowptr, self.num_nodes, threshold, segment)
        return self._hubs[key]

    def transposed_hub_tables(self, threshold=HUB_THRESHOLD, segment=HUB_SEGMENT):
        """the same over the by-SOURCE view (`transposed()`): sources with >= `threshold` out-edges; offsets into t_eid / t_dst"""
        key = ("t", int(threshold), int(segment))
        if key not in self._hubs:
            self._hubs[key] = DstCSR._hub_tables_of(self.transposed()[0], self.num_nodes, threshold, segment)
        return self._hubs[key]

    def tile_need(self, mask_u8, rows_per_tile=32, table_mask_u8=None):
        """Per 32-row tile: which of a node's two transformed rows does the aggregation over THIS graph ever read?  bit 0 = h_s2t
        (gathered by target-domain destinations, KTGNN.py:293,:295), bit 1 = h_t2s (source-domain destinations, :292,:294); a
        row's own table counts (the logit reads h_i).  -> int32 [ceil(rows / 32)] for `adaptedconv_transform(tile_need=...)`, or
        None when every tile needs both tables.  Built once per (graph, mask); with s -> t bridge edges only, no target node
        feeds a source destination and the target half of h_t2s is never read.
        `table_mask_u8` (a rank's graph: destinations = its own rows, tables = own rows followed by halo rows): the domain flags of
        ALL table rows; the tiles then cover the extended tables and a halo row counts only where an edge reads it."""
        key = (mask_u8.data_ptr(), mask_u8._version, int(rows_per_tile), None if table_mask_u8 is None else table_mask_u8.data_ptr())
        c = getattr(self, "_tile_need", None)
        if c is None or c[0] != key:
            E, N = self.num_edges, self.num_nodes
            R = N if table_mask_u8 is None else int(table_mask_u8.shape[0])      # rows of the tables
            m = mask_u8[:N].bool()
            deg = (self.rowptr[1:N + 1] - self.rowptr[:N]).long()
            dst_s = torch.repeat_interleave(m, deg)                       # domain of every edge's destination (by-destination order)
            col = self.col[:E].long()
            need_t2s = torch.zeros(R, dtype=torch.bool, device=m.device)
            need_s2t = torch.zeros(R, dtype=torch.bool, device=m.device)
            need_t2s[:N], need_s2t[:N] = m, ~m
            need_t2s[col[dst_s]] = True
            need_s2t[col[~dst_s]] = True
            T = (R + rows_per_tile - 1) // rows_per_tile
            pad = T * rows_per_tile - R

            def tiles(v):
                v = torch.cat((v, v.new_zeros(pad))) if pad else v
                return v.view(T, rows_per_tile).any(1)
            need = (tiles(need_s2t).to(torch.int32) | (tiles(need_t2s).to(torch.int32) << 1)).contiguous()
            full = bool((need == 3).all().item())                         # one-time sync, like the CSR build
            self._tile_need = c = (key, None if full else need, mask_u8, table_mask_u8)  # (masks kept alive: the key holds their addresses)
        return c[1]

    def gather_hint(self, window=1024, samples=64):
        """1: neighbouring destination rows share neighbours (the aggregation's gathers live off the L2s), 2: they do not (HBM-bound
        gathers) -- `bgnn_adaptedconv_aggregate_bounded_f32(gather_hint=...)`.  Measured once per graph: over `samples` evenly spaced
        windows of `window` consecutive destination rows (about what one XCD has in flight), reuse = 1 - distinct neighbour ids / edges."""
        if getattr(self, "_gather_hint", None) is None:
            N, E = self.num_nodes, self.num_edges
            hint = 1
            if N > 4 * window and E > 0:
                starts = torch.linspace(0, N - window, samples, device=self.rowptr.device).long()
                rp = self.rowptr.long()
                tot = dist = 0
                b, e = rp[starts].tolist(), rp[starts + window].tolist()            # one-time sync, like tile_need / the CSR build
                for lo, hi in zip(b, e):
                    if hi > lo:
                        tot += hi - lo
                        dist += int(torch.unique(self.col[lo:hi]).numel())
                if tot > 0 and 1.0 - dist / tot < 0.3:
                    hint = 2
            self._gather_hint = hint
        return self._gather_hint

    def transposed(self):
        """By-SOURCE view of the same edges, built once per graph (training only): (t_rowptr [N+1], t_eid [E'] = position
        of the edge in the by-destination order, t_dst [E'] = its destination), int32.  The atomic-free aggregation
        backward gathers over it instead of scattering with float atomics."""
        if self._transposed is None:
            E, N = self.num_edges, self.num_nodes
            col = self.col[:E].long()
            deg_in = (self.rowptr[1:] - self.rowptr[:-1]).long()
            dst = torch.repeat_interleave(torch.arange(N, device=col.device), deg_in)
            order = torch.sort(col, stable=True).indices
            t_rowptr = torch.zeros(N + 1, dtype=torch.int64, device=col.device)
            t_rowptr[1:] = torch.cumsum(torch.bincount(col, minlength=N), 0)
            self._transposed = (t_rowptr.to(torch.int32).contiguous(), order.to(torch.int32).contiguous(),
                                dst[order].to(torch.int32).contiguous())
        return self._transposed


def build_dst_csr(edge_index, num_nodes, rewrite_self_loops=True, want_eperm=False):
    """edge_index int64 [2,E] (CUDA) -> DstCSR.  One D2H read of E' (the CSR is built once per graph
    and cached, like the reference caches graph_partition)."""
    lib = L.lib()
    if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError("edge_index must be int64 [2, E]")
    ei = edge_index.contiguous()
    E, N = int(ei.shape[1]), int(num_nodes)
    dev = ei.device
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    col = torch.empty(E + N, dtype=torch.int32, device=dev)
    eperm = torch.empty(E + N, dtype=torch.int32, device=dev) if want_eperm else None
    e_out = torch.zeros(1, dtype=torch.int64, device=dev)
    wsb = lib.bgnn_csr_workspace_bytes(N, E)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    rc = lib.bgnn_build_dst_csr(L.ptr(ei) if E > 0 else None, E, N, 1 if rewrite_self_loops else 0, L.ptr(rowptr),
                                L.ptr(col), L.ptr(eperm), L.ptr(e_out), L.ptr(ws), wsb, L.stream())
    L.check(rc, "bgnn_build_dst_csr")
    ne = int(e_out.item())
    return DstCSR(rowptr, col[:ne], eperm[:ne] if want_eperm else None, ne, N)


class ZeroArena:
    """One zero-filled float64 allocation per forward that hands out the [2*Din+2] accumulators of `domain_sums` and of
    the `colsum` epilogues (one fill launch instead of one per accumulator)."""

    def __init__(self, device, doubles):
        self.buf = torch.zeros(int(doubles), dtype=torch.float64, device=device)
        self.used = 0

    def take(self, n):
        n = int(n)
        if self.used + n > self.buf.numel():            # undersized arena: still correct, one more fill
            return torch.zeros(n, dtype=torch.float64, device=self.buf.device)
        t = self.buf[self.used: self.used + n]
        self.used += (n + 1) // 2 * 2                   # keep slices 16-byte aligned
        return t


def domain_sums(x, mask_u8, out=None, deterministic=False):
    """Per-domain column sums + counts as a float64 [2*Din+2] tensor (all-reducible).  `out`: zero-filled accumulator.
    `deterministic`: the two-stage form (per-block partial rows + a second small launch, no atomics: run-to-run
    bit-identical); measured the same speed at C4 size and no faster on a rank's share, so the one-launch form stays
    the default."""
    lib = L.lib()
    N, Din = x.shape
    sums = torch.zeros(2 * Din + 2, dtype=torch.float64, device=x.device) if out is None else out
    assert sums.numel() == 2 * Din + 2 and sums.dtype == torch.float64
    if not deterministic:
        rc = lib.bgnn_domain_sums_f64(L.ptr(x), N, Din, x.stride(0), L.ptr(mask_u8), L.ptr(sums), L.stream())
        L.check(rc, "bgnn_domain_sums_f64")
        return sums
    ws = torch.empty(lib.bgnn_domain_sums_workspace_bytes(Din), dtype=torch.uint8, device=x.device)   # per call (see _tile_queue)
    rc = lib.bgnn_domain_sums_ws_f64(L.ptr(x), N, Din, x.stride(0), L.ptr(mask_u8), L.ptr(sums), L.ptr(ws), ws.numel(), L.stream())
    L.check(rc, "bgnn_domain_sums_ws_f64")
    return sums


def column_sums(x):
    """sum over the rows of x [N, D] (fp64 accumulation, D % 4 == 0 and a 16-byte aligned, row-contiguous x) -> float32 [D]: the bias
    gradient of a Linear.  One streaming launch of the domain-sums kernel with every row in one domain; torch's `x.sum(0)` is a
    multi-block reduction whose semaphore buffer is cleared by a memset node when captured into a HIP graph (see bgnn_zero_async)."""
    N, D = x.shape
    mask = torch.zeros(N, dtype=torch.uint8, device=x.device)
    return domain_sums(x, mask)[D:2 * D].float()


def total_sum(x):
    """sum of all elements of a float32 tensor as a 0-dim float32 tensor, through `column_sums` (no torch multi-block reduction: safe
    inside a captured training step, see `KTGNN_no_complement.graphed_train_step`).  Differentiable (d/dx = 1)."""
    return _TotalSumFn.apply(x)


class _TotalSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        flat = x.contiguous().view(-1)
        pad = (-flat.numel()) % 64
        if pad:
            flat = torch.cat((flat, flat.new_zeros(pad)))
        return column_sums(flat.view(-1, 64)).sum()          # 64 values: a single-block reduction

    @staticmethod
    def backward(ctx, g):
        return g.expand(ctx.shape)


def domain_delta(sums, Din):
    delta = torch.empty(Din, dtype=torch.float32, device=sums.device)
    rc = L.lib().bgnn_domain_delta_f32(L.ptr(sums), Din, L.ptr(delta), L.stream())
    L.check(rc, "bgnn_domain_delta_f32")
    return delta


def linear_supported(din, dout):
    """envelope of `linear` (the W-stationary MFMA kernel): Din <= 128, Din % 4 == 0, Dout % 64 == 0."""
    return din <= 128 and din % 4 == 0 and dout % 64 == 0


def linear(x, weight, bias, relu=False, mask_u8=None, colsum=None):
    """relu?(x @ weight.T + bias) (KTGNN.py:407-411/:433, clf_transformer's first Linear with eval BatchNorm folded in);
    `colsum` (zeroed float64 [2*Dout+2], needs mask_u8) also receives the per-domain column sums of the result."""
    N, din = x.shape
    dout = weight.shape[0]
    out = torch.empty(N, dout, dtype=torch.float32, device=x.device)
    rc = L.lib().bgnn_linear_f32(L.ptr_rows(x), N, din, x.stride(0), L.ptr(weight), L.ptr(bias), dout, 1 if relu else 0,
                                 L.ptr(mask_u8), L.ptr(colsum), L.ptr(out), dout, L.stream())
    L.check(rc, "bgnn_linear_f32")
    return out


def linear_narrow_supported(din, dout, packed):
    """envelope of `linear_narrow_transform`: W-stationary kernel with the whole activation row in one column group, and a
    consumer conv of one head with D <= 4."""
    Wp, bp, gates, D, ldh, gconst = packed
    return (din <= 128 and din % 4 == 0 and dout in (64, 128, 256) and gates.shape[0] == 1 and ldh == 4
            and Wp.shape == (8, dout))


def linear_narrow_transform(x, weight, bias, mask_u8, colsum, packed, relu=True):
    """Stage A of the fused clf_transformer -> clf_target path (KTGNN.py:433): raw [N,12] per-row products of
    a = relu?(x W^T + b) with the consumer conv's packed rows and gate vectors, `colsum` (zeroed float64 [2*Dout+2])
    += the per-domain column sums of a.  The activation itself is never written."""
    N, din = x.shape
    dout = weight.shape[0]
    Wp, bp, gates, D, ldh, gconst = packed
    raw = torch.empty(N, 12, dtype=torch.float32, device=x.device)
    rc = L.lib().bgnn_linear_narrow_transform_f32(L.ptr_rows(x), N, din, x.stride(0), L.ptr(weight), L.ptr(bias), dout,
                                                  1 if relu else 0, L.ptr(mask_u8), L.ptr(colsum), L.ptr(Wp), L.ptr(gates),
                                                  L.ptr(raw), L.stream())
    L.check(rc, "bgnn_linear_narrow_transform_f32")
    return raw


def classifier_stage_supported(x, packed_pair, weight, packed_t):
    """envelope of `classifier_stage` (bgnn.h: bgnn_classifier_stage_f32)"""
    Wp, bp, gates, D, ldh, gconst = packed_pair
    din = x.shape[1]
    return (x.is_cuda and x.dtype == torch.float32 and x.stride(1) == 1 and 64 < din <= 128 and din % 4 == 0
            and weight.shape == (128, din) and Wp.shape[0] <= 24 and Wp.shape[1] == din and packed_t[0].shape == (8, 128)
            and os.environ.get("BGNN_FUSED_CLS", "1") != "0")


def classifier_stage(x, mask_u8, sums_x, packed_pair, outs, weight, bias, colsum, packed_t, relu=True):
    """ONE pass over the hidden activation x for the whole classifier stage's dense work (KTGNN.py:432-434): the narrow
    (h_t2s, h_s2t) tables of the convs in `packed_pair` (clf_base, clf_target on x; `outs` = list of (h_t2s, h_s2t) views) and
    stage A of the fused clf_transformer -> clf_target path (`linear_narrow_transform`: -> raw [N, 12], `colsum` += the
    per-domain column sums of the activation)."""
    N, din = x.shape
    Wp, bp, gates, D, ldh, gconst = packed_pair
    H = gates.shape[0]
    Wp2, _, gates2, _, _, _ = packed_t
    raw = torch.empty(N, 12, dtype=torch.float32, device=x.device)
    small = torch.empty(H * (2 * ldh + 2) + 8, dtype=torch.float32, device=x.device)
    row_stride = outs[0][0].stride(0)
    for a, b in outs:
        assert a.shape[0] >= N and b.shape[0] >= N and a.stride(0) == row_stride and b.stride(0) == row_stride
    o1 = outs[1] if H > 1 else (None, None)
    rc = L.lib().bgnn_classifier_stage_f32(L.ptr_rows(x), N, din, x.stride(0), L.ptr(mask_u8), L.ptr(sums_x), H, D, L.ptr(Wp), L.ptr(bp),
                                           L.ptr(gates), L.ptr(gconst), L.ptr_rows(outs[0][1]), L.ptr_rows(outs[0][0]),
                                           L.ptr_rows(o1[1]), L.ptr_rows(o1[0]), ldh, row_stride, L.ptr(weight), L.ptr(bias),
                                           weight.shape[0], 1 if relu else 0, L.ptr(colsum), L.ptr(Wp2), L.ptr(gates2), L.ptr(raw),
                                           L.ptr(small), L.stream())
    L.check(rc, "bgnn_classifier_stage_f32")
    return raw


def narrow_transform_finish(raw, mask_u8, sums, packed, out):
    """Stage B: raw [N,12] + the (all-reduced) domain sums of the activation -> the conv's (h_t2s, h_s2t) rows in `out`
    (two [>=N, 4] views with a common row stride)."""
    Wp, bp, gates, D, ldh, gconst = packed
    h_t2s, h_s2t = out
    N = raw.shape[0]
    assert h_t2s.stride(0) == h_s2t.stride(0) and sums.dtype == torch.float64 and sums.numel() == 2 * Wp.shape[1] + 2
    small = torch.empty(16, dtype=torch.float32, device=raw.device)
    rc = L.lib().bgnn_narrow_transform_finish_f32(L.ptr(raw), N, L.ptr(mask_u8), L.ptr(sums), Wp.shape[1], L.ptr(Wp), L.ptr(bp),
                                                  L.ptr(gates), L.ptr(gconst), L.ptr_rows(h_s2t), L.ptr_rows(h_t2s),
                                                  h_t2s.stride(0), L.ptr(small), L.stream())
    L.check(rc, "bgnn_narrow_transform_finish_f32")
    return out


def transform_bwd_prep(x, G_s2t, G_t2s, D, mask_u8, gx, gconst, wd, counts, out=None, want_ex=False):
    """-> (Gall [N, pad4(2D+3)], side [N, 4]): the row-local part of the transform backward in one pass (see bgnn.h);
    `counts` = float64 [2] (n_S, n_T), e.g. the tail of the domain sums.  `out` = (Gall view, side view or None): column
    slices of wider buffers when several convs on the same x share the launches that follow.
    want_ex (D <= 128): -> (Gall, ex [p, 4]) instead -- the used entries of Gall^T side from the same pass, no side buffer."""
    N, din = x.shape
    p = pad4(2 * D + 3)
    assert counts.dtype == torch.float64 and counts.numel() == 2
    if out is None:
        Gall = torch.empty(N, p, dtype=torch.float32, device=x.device)
        side = None if want_ex else torch.empty(N, 4, dtype=torch.float32, device=x.device)
    else:
        Gall, side = out
        assert Gall.shape == (N, p) and (side is None or side.shape == (N, 4))
    assert G_s2t.stride(0) == G_t2s.stride(0) and G_s2t.stride(1) == 1 and G_t2s.stride(1) == 1
    lib = L.lib()
    ex = ws = None
    wsb = 0
    if want_ex:
        ex = torch.empty(p, 4, dtype=torch.float32, device=x.device)
        wsb = lib.bgnn_transform_bwd_prep_workspace_bytes(N, p)
        ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    rc = lib.bgnn_transform_bwd_prep_f32(L.ptr_rows(x), x.stride(0), N, din, L.ptr_rows(G_s2t), L.ptr_rows(G_t2s),
                                         G_s2t.stride(0), D, L.ptr(mask_u8), L.ptr(gx), L.ptr(gconst), L.ptr(wd),
                                         L.ptr(counts), L.ptr_rows(Gall), p, Gall.stride(0),
                                         L.ptr_rows(side) if side is not None else None, side.stride(0) if side is not None else 4,
                                         L.ptr(ex), L.ptr(ws), wsb, L.stream())
    L.check(rc, "bgnn_transform_bwd_prep_f32")
    return (Gall, ex) if want_ex else (Gall, side)


def transform_bwd_consts(W_s, W_t, g1, g2, delta, din):
    """-> (gx [2, din], gconst [2], wd [2, 2D]): the small operands of `transform_bwd_prep` in one launch (see bgnn.h)."""
    D = W_s.shape[0]
    dev = W_s.device
    gx = torch.empty(2, din, dtype=torch.float32, device=dev)
    gconst = torch.empty(2, dtype=torch.float32, device=dev)
    wd = torch.empty(2, 2 * D, dtype=torch.float32, device=dev)
    rc = L.lib().bgnn_transform_bwd_consts_f32(L.ptr(W_s), L.ptr(W_t), L.ptr(g1), L.ptr(g2), L.ptr(delta), D, din,
                                               L.ptr(gx), L.ptr(gconst), L.ptr(wd), L.stream())
    L.check(rc, "bgnn_transform_bwd_consts_f32")
    return gx, gconst, wd


def transform_bwd_finish(dWall, ex, W_s, W_t, g1, g2, delta, din, wcat_t):
    """-> (dW_s, dW_t, dg1 [2 din], dg2 [2 din], db_s, db_t); fills wcat_t ([din, p] view, unit column stride): see bgnn.h."""
    D = W_s.shape[0]
    p = dWall.shape[0]
    dev = W_s.device
    dW_s, dW_t = torch.empty(D, din, dtype=torch.float32, device=dev), torch.empty(D, din, dtype=torch.float32, device=dev)
    dg1, dg2 = torch.empty(2 * din, dtype=torch.float32, device=dev), torch.empty(2 * din, dtype=torch.float32, device=dev)
    db_s, db_t = torch.empty(D, dtype=torch.float32, device=dev), torch.empty(D, dtype=torch.float32, device=dev)
    assert dWall.is_contiguous() and ex.is_contiguous() and wcat_t.shape == (din, p) and wcat_t.stride(1) == 1
    rc = L.lib().bgnn_transform_bwd_finish_f32(L.ptr(dWall), L.ptr(ex), L.ptr(W_s), L.ptr(W_t), L.ptr(g1), L.ptr(g2), L.ptr(delta),
                                               D, din, p, L.ptr(dW_s), L.ptr(dW_t), L.ptr(dg1), L.ptr(dg2), L.ptr(db_s), L.ptr(db_t),
                                               L.ptr_rows(wcat_t), wcat_t.stride(0), L.stream())
    L.check(rc, "bgnn_transform_bwd_finish_f32")
    return dW_s, dW_t, dg1, dg2, db_s, db_t


def gram_supported(p, q):
    return 0 < p <= 288 and 0 < q <= 128 and p % 4 == 0 and q % 4 == 0


def gram(A, B):
    """A^T B for tall-skinny row-major A [N,p], B [N,q] (the node count is the reduction dimension): the weight-gradient
    products of the training path (KTGNN.py:275-284 under autograd)."""
    N, p = A.shape
    q = B.shape[1]
    lib = L.lib()
    wsb = lib.bgnn_gram_workspace_bytes(p, q)
    ws = torch.empty(wsb, dtype=torch.uint8, device=A.device)
    out = torch.empty(p, q, dtype=torch.float32, device=A.device)
    rc = lib.bgnn_gram_f32(L.ptr_rows(A), A.stride(0), p, L.ptr_rows(B), B.stride(0), q, N, L.ptr(out), L.ptr(ws), wsb, L.stream())
    L.check(rc, "bgnn_gram_f32")
    return out


def bn_relu_dropout_supported(x):
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1 and x.shape[1] % 4 == 0
            and 4 <= x.shape[1] <= 1024 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)


def bn_relu_dropout(x, gamma, beta, eps, relu, p_drop, seed, momentum=0.0, running_mean=None, running_var=None, seed_dev=None):
    """Training-mode BatchNorm1d -> ReLU -> dropout (KTGNN.py:420-430) -> (y, stats); `stats` (fp64 column sums of x | x^2)
    is what the backward needs besides x.  `seed_dev`: optional int64 device tensor [1] added to `seed` inside the kernels (the step
    counter of a captured training step)."""
    N, D = x.shape
    y = torch.empty(N, D, dtype=torch.float32, device=x.device)
    stats = torch.empty(L.lib().bgnn_bn_acc_doubles(D), dtype=torch.float64, device=x.device)   # R x [2D] partial accumulators
    rc = L.lib().bgnn_bn_relu_dropout_f32(L.ptr_rows(x), N, D, x.stride(0), L.ptr(gamma) if gamma is not None else None,
                                          L.ptr(beta) if beta is not None else None, float(eps), int(bool(relu)), float(p_drop),
                                          int(seed) & 0xFFFFFFFFFFFFFFFF, L.ptr(seed_dev) if seed_dev is not None else None, float(momentum),
                                          L.ptr(running_mean) if running_mean is not None else None,
                                          L.ptr(running_var) if running_var is not None else None,
                                          L.ptr(y), D, L.ptr(stats), L.stream())
    L.check(rc, "bgnn_bn_relu_dropout_f32")
    return y, stats


def bn_relu_dropout_bwd(x, grad_y, stats, gamma, beta, eps, relu, p_drop, seed, seed_dev=None):
    """-> (dL/dx [N,D], gsum fp64 [2*D] = dL/dbeta | dL/dgamma)."""
    N, D = x.shape
    gx = torch.empty(N, D, dtype=torch.float32, device=x.device)
    gsum = torch.empty(L.lib().bgnn_bn_acc_doubles(D), dtype=torch.float64, device=x.device)
    rc = L.lib().bgnn_bn_relu_dropout_bwd_f32(L.ptr_rows(x), L.ptr_rows(grad_y), N, D, x.stride(0), grad_y.stride(0), L.ptr(stats),
                                              L.ptr(gamma) if gamma is not None else None, L.ptr(beta) if beta is not None else None,
                                              float(eps), int(bool(relu)), float(p_drop), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                              L.ptr(seed_dev) if seed_dev is not None else None,
                                              L.ptr(gx), D, L.ptr(gsum), L.stream())
    L.check(rc, "bgnn_bn_relu_dropout_bwd_f32")
    return gx, gsum.view(-1, 2 * D).sum(0)


def rowdot(X, V):
    """X [N,d] (unit column stride, 16-B aligned rows) times up to four vectors V [nv,d] -> [N,nv] in one pass over X."""
    N, d = X.shape
    nv = V.shape[0]
    out = torch.empty(N, nv, dtype=torch.float32, device=X.device)
    rc = L.lib().bgnn_rowdot_f32(L.ptr_rows(X), X.stride(0), N, d, L.ptr(V), V.stride(0), nv, L.ptr(out), L.stream())
    L.check(rc, "bgnn_rowdot_f32")
    return out


def pack_transform_heads(heads, din_pad):
    """Pack 1-2 convs that share an input for `adaptedconv_transform`.  `heads` = list of dicts with
    W_s, b_s, W_t, b_t ([D, Din] / [D] or None) and g_s2t, g_t2s ([2*Din], [x || delta] order).
    Optional per head: "gate_const" = (c_s2t, c_t2s) added to the gate pre-activations.
    -> (Wp [H*2*ldh, din_pad], bias_p [H*2*ldh], gates [H, 2, 2*din_pad], D, ldh, gate_const [H, 2])."""
    D = heads[0]["W_s"].shape[0]
    din = heads[0]["W_s"].shape[1]
    ldh = pad4(D)
    dev = heads[0]["W_s"].device
    H = len(heads)
    Wp = torch.zeros(H * 2 * ldh, din_pad, dtype=torch.float32, device=dev)
    bp = torch.zeros(H * 2 * ldh, dtype=torch.float32, device=dev)
    gates = torch.zeros(H, 2, 2 * din_pad, dtype=torch.float32, device=dev)
    gconst = torch.zeros(H, 2, dtype=torch.float32, device=dev)
    for h, hd in enumerate(heads):
        if hd.get("gate_const") is not None:
            gconst[h, 0], gconst[h, 1] = hd["gate_const"]
        base = h * 2 * ldh
        Wp[base: base + D, :din] = hd["W_t"]
        Wp[base + ldh: base + ldh + D, :din] = hd["W_s"]
        if hd.get("b_t") is not None:
            bp[base: base + D] = hd["b_t"]
        if hd.get("b_s") is not None:
            bp[base + ldh: base + ldh + D] = hd["b_s"]
        for t, key in enumerate(("g_s2t", "g_t2s")):
            g = hd[key].reshape(-1)
            gates[h, t, :din] = g[:din]
            gates[h, t, din_pad: din_pad + din] = g[din:]
    return Wp, bp, gates, D, ldh, gconst


def adaptedconv_transform(x, mask_u8, delta, packed, out=None, sums=None, tail_single=(0, 0), tile_need=None):
    """One pass over x -> per head (h_t2s, h_s2t) as [N, ldh] tensors (ldh = pad4(D); columns >= D are
    zero; `tile_need`: see DstCSR.tile_need).  `packed` = pack_transform_heads(...).  `out` = list of (h_t2s, h_s2t) preallocated tables
    (>= N rows, row stride ldh; multi-GPU: halo rows follow the N local rows).  With `delta=None` the domain
    `sums` ([2*Din+2] float64) are consumed directly (same delta, one launch less).  `tail_single=(n_t2s, n_s2t)`
    (sums form only): the last n_t2s + n_s2t rows need only h_t2s / only h_s2t; their other table may stay unwritten."""
    lib = L.lib()
    Wp, bp, gates, D, ldh, gconst = packed
    H = gates.shape[0]
    N, Din = x.shape
    dev = x.device
    if out is None:
        out = [(torch.empty(N, ldh, dtype=torch.float32, device=dev), torch.empty(N, ldh, dtype=torch.float32, device=dev))
               for _ in range(H)]
    row_stride = out[0][0].stride(0)
    for a, b in out:
        assert a.shape[0] >= N and b.shape[0] >= N and a.stride(0) == row_stride and b.stride(0) == row_stride
    small = torch.empty(H * (2 * ldh + 2) + 8, dtype=torch.float32, device=dev)
    o1 = out[1] if H > 1 else (None, None)
    if delta is None:
        if sums is None or sums.dtype != torch.float64 or sums.numel() != 2 * Din + 2:
            raise ValueError("adaptedconv_transform needs delta [Din] or the float64 domain sums [2*Din+2]")
        fn, first, name = lib.bgnn_adaptedconv_transform_sums_f32, sums, "bgnn_adaptedconv_transform_sums_f32"
    else:
        fn, first, name = lib.bgnn_adaptedconv_transform_f32, delta, "bgnn_adaptedconv_transform_f32"
    tail = (int(tail_single[0]), int(tail_single[1])) if delta is None else ()
    if delta is not None and tuple(tail_single) != (0, 0):
        raise ValueError("tail_single needs the sums form of the transform")
    if tile_need is not None:
        # `tile_need` (DstCSR.tile_need): rows of a table that the aggregation never reads may stay unwritten (sums form only)
        if delta is not None or tuple(tail_single) != (0, 0):
            raise ValueError("tile_need needs the sums form of the transform and no tail_single")
        if tile_need.dtype != torch.int32 or tile_need.numel() != (N + 31) // 32:
            raise ValueError("tile_need: one int32 per 32-row tile")
        fn, name, tail = lib.bgnn_adaptedconv_transform_need_f32, "bgnn_adaptedconv_transform_need_f32", (L.ptr(tile_need),)
    rc = fn(L.ptr(x), N, Din, x.stride(0), L.ptr(mask_u8), L.ptr(first), H, D, L.ptr(Wp), L.ptr(bp), L.ptr(gates), L.ptr(gconst),
            L.ptr_rows(out[0][1]), L.ptr_rows(out[0][0]), L.ptr_rows(o1[1]), L.ptr_rows(o1[0]), ldh, row_stride, *tail,
            L.ptr(small), L.stream())
    L.check(rc, name)
    return out


# second-part launches (a boundary row's two or three remote-source edges) run without the tile queue: nothing to keep
# L2-resident there, and the claims' latency is most of such a short row's time (0.67 -> 0.64 ms per rank-sized forward)
_P2_STATIC = os.environ.get("BGNN_P2_STATIC", "1") != "0"


def _tile_queue(dev):
    """8 x uint32 scratch for the aggregation kernel's per-XCD dynamic tile counters (the C entry zeroes it on the stream per
    launch, with a kernel of its own -- see bgnn_zero_async in csrc/bgnn_common.h).  A fresh allocation per call: the scratch has
    no life outside its launch, so nothing created inside a HIP-graph capture has to stay valid between replays."""
    return torch.empty(8, dtype=torch.int32, device=dev)


def heads_log_softmax_supported(heads, D):
    """Envelope of the fused log_softmax epilogue (bgnn.h: ep_relu == 2)."""
    return heads in (2, 3) and D <= 4


def _hub_shape_ok(D, ldh, ldo, heads, slope):
    """the shapes whose kernels understand hub segments (bgnn.h: bgnn_adaptedconv_aggregate_hub_f32)"""
    if heads in (2, 3) and D <= 4 and ldh == 4 and ldo == 4:
        return True
    return heads == 1 and D > 32 and 0.0 <= slope <= 1.0


def adaptedconv_aggregate(h_t2s, h_s2t, a_t2s, a_s2t, csr, mask_u8, D, negative_slope=0.1, n_dst=None,
                          want_alpha=False, ep_scale=None, ep_shift=None, ep_relu=False, out=None,
                          row_begin=0, row_end=None, state_ms=None, part=0, heads=1, colsum=None, log_softmax=False,
                          park_begin=None):
    """-> out [n_dst, pad4(D)] (use out[:, :D]); optionally alpha [E'] in CSR order.
    part=1: rows [row_begin, park_begin) are finished in this launch, rows [park_begin, row_end) parked for part=2
    (default: all parked).
    `log_softmax` (interleaved narrow heads only, see `heads_log_softmax_supported`): the finished rows leave the kernel
    as log_softmax over each head's D classes (KTGNN.py:435).
    Only rows [row_begin, row_end) are computed (default: all n_dst rows).
    heads > 1: tables are [rows, heads*pad4(D)] (heads interleaved per node), a_* are [heads, D], out is
    [n_dst, heads*pad4(D)]: one pass over the CSR serves all heads."""
    lib = L.lib()
    n_dst = csr.num_nodes if n_dst is None else int(n_dst)
    row_end = n_dst if row_end is None else int(row_end)
    ldh = h_t2s.stride(0) // heads
    ldo = pad4(D)
    dev = h_t2s.device
    if out is None:
        out = torch.empty(n_dst, heads * ldo, dtype=torch.float32, device=dev)
    assert h_t2s.stride(0) == h_s2t.stride(0) and h_t2s.stride(0) % heads == 0 and out.stride(0) % heads == 0
    alpha = torch.empty(csr.num_edges, dtype=torch.float32, device=dev) if want_alpha else None
    tq = _tile_queue(dev) if heads == 1 else None            # (alive until the launch below has been issued)
    if row_end <= int(row_begin):                    # empty row range (e.g. no boundary rows at world size 1)
        return (out, alpha) if want_alpha else out
    # graphs with hub rows: segments + merge (bgnn.h).  Whole-graph, single-launch calls of the two hub-aware kernels only.
    if ((not want_alpha or heads == 1) and part in (0, 3) and int(row_begin) == 0 and row_end == csr.num_nodes == n_dst and h_t2s.shape[0] == n_dst
            and (ep_scale is None or heads == 1) and _hub_shape_ok(D, ldh, out.stride(0) // heads, heads, negative_slope)
            and os.environ.get("BGNN_HUB_ROWS", "1") != "0"):
        hubs = csr.hub_tables()
        if hubs is not None:
            hub_rows, seg_ptr, seg_bounds, seg_node = hubs
            nseg = int(seg_node.numel())
            wsb = lib.bgnn_aggregate_hub_workspace_bytes(nseg, heads, out.stride(0) // heads)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            rc = lib.bgnn_adaptedconv_aggregate_hub_f32(
                L.ptr_rows(h_t2s), L.ptr_rows(h_s2t), ldh, L.ptr(a_t2s), L.ptr(a_s2t), L.ptr(csr.rowptr), L.ptr(csr.col),
                L.ptr(mask_u8), n_dst, D, float(negative_slope), L.ptr_rows(out), out.stride(0) // heads,
                L.ptr(ep_scale), L.ptr(ep_shift), 2 if log_softmax else (1 if ep_relu else 0),
                L.ptr(state_ms) if part == 3 else None, int(heads), L.ptr(colsum),
                L.ptr(tq), HUB_THRESHOLD, L.ptr(hub_rows), int(hub_rows.numel()),
                L.ptr(seg_ptr), L.ptr(seg_bounds), L.ptr(seg_node), nseg, L.ptr(alpha), L.ptr(ws), wsb, L.stream())
            L.check(rc, "bgnn_adaptedconv_aggregate_hub_f32")
            return (out, alpha) if want_alpha else out
    # the tables' row count bounds every id in csr.col (a CSR over these tables): lets the plain wide launch use 32-bit addressing
    rc = lib.bgnn_adaptedconv_aggregate_bounded_f32(
        L.ptr_rows(h_t2s), L.ptr_rows(h_s2t), ldh, L.ptr(a_t2s), L.ptr(a_s2t), L.ptr(csr.rowptr), L.ptr(csr.col),
        L.ptr(mask_u8), int(row_begin), row_end, D, float(negative_slope), L.ptr_rows(out), out.stride(0) // heads, L.ptr(alpha),
        L.ptr(ep_scale), L.ptr(ep_shift), 2 if log_softmax else (1 if ep_relu else 0), L.ptr(state_ms), int(part),
        int(row_begin) if park_begin is None else int(park_begin), int(heads), L.ptr(colsum),
        L.ptr(tq) if not (part == 2 and _P2_STATIC) else None, min(int(h_t2s.shape[0]), int(h_s2t.shape[0])),
        csr.gather_hint() if (heads == 1 and D > 32 and part == 0) else 0, L.stream())
    L.check(rc, "bgnn_adaptedconv_aggregate_bounded_f32")
    return (out, alpha) if want_alpha else out


def adaptedconv_aggregate_bwd(h_t2s, h_s2t, a_t2s, a_s2t, csr, mask_u8, D, out, alpha, grad_out, negative_slope=0.1):
    """-> (dh_t2s, dh_s2t, da_t2s, da_s2t): gradients of the fused aggregation w.r.t. both tables and
    both attention vectors (reference: autograd through models/KTGNN.py:292-305)."""
    lib = L.lib()
    dev = h_t2s.device
    da_t2s = torch.zeros(D, dtype=torch.float32, device=dev)
    da_s2t = torch.zeros(D, dtype=torch.float32, device=dev)
    grad_out = grad_out.contiguous()
    if D <= 128 and grad_out.stride(0) % 4 == 0 and grad_out.data_ptr() % 16 == 0 and h_t2s.shape[0] == csr.num_nodes:
        # atomic-free pull over the by-source CSR (float atomics retire at ~1.3 TB/s on MI355X); the atomic form remains for
        # D > 128 and for row ranges
        t_rowptr, t_eid, t_dst = csr.transposed()
        dh_t2s, dh_s2t = torch.empty_like(h_t2s), torch.empty_like(h_s2t)
        narrow = D <= 4 and h_t2s.stride(0) == 4 and out.stride(0) == 4 and grad_out.stride(0) == 4
        dh_, sh_ = (csr.hub_tables(), csr.transposed_hub_tables()) if (not narrow and os.environ.get("BGNN_HUB_ROWS", "1") != "0") else (None, None)
        if dh_ is not None or sh_ is not None:      # graphs with hub rows: segments + merge (bgnn.h)
            none4 = (None, None, None, None)
            d_rows, d_ptr, d_bounds, d_node = dh_ if dh_ is not None else none4
            s_rows, s_ptr, s_bounds, s_node = sh_ if sh_ is not None else none4
            nd = 0 if dh_ is None else int(d_node.numel())
            ns = 0 if sh_ is None else int(s_node.numel())
            wsb = lib.bgnn_aggregate_bwd_pull_hub_workspace_bytes(csr.num_nodes, csr.num_edges, h_t2s.stride(0), nd, ns)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            rc = lib.bgnn_adaptedconv_aggregate_bwd_pull_hub_f32(
                L.ptr(h_t2s), L.ptr(h_s2t), h_t2s.stride(0), L.ptr(a_t2s), L.ptr(a_s2t), L.ptr(csr.rowptr), L.ptr(csr.col),
                L.ptr(mask_u8), L.ptr(t_rowptr), L.ptr(t_eid), L.ptr(t_dst), csr.num_nodes, csr.num_edges, D, float(negative_slope),
                L.ptr(out), out.stride(0), L.ptr(alpha), L.ptr(grad_out), grad_out.stride(0),
                L.ptr(dh_t2s), L.ptr(dh_s2t), L.ptr(da_t2s), L.ptr(da_s2t), HUB_THRESHOLD,
                L.ptr(d_rows), 0 if dh_ is None else int(d_rows.numel()), L.ptr(d_ptr), L.ptr(d_bounds), L.ptr(d_node), nd,
                L.ptr(s_rows), 0 if sh_ is None else int(s_rows.numel()), L.ptr(s_ptr), L.ptr(s_bounds), L.ptr(s_node), ns,
                L.ptr(ws), wsb, L.stream())
            L.check(rc, "bgnn_adaptedconv_aggregate_bwd_pull_hub_f32")
            return dh_t2s, dh_s2t, da_t2s, da_s2t
        wsb = lib.bgnn_aggregate_bwd_pull_workspace_bytes(csr.num_nodes, csr.num_edges, h_t2s.stride(0))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        rc = lib.bgnn_adaptedconv_aggregate_bwd_pull_f32(
            L.ptr(h_t2s), L.ptr(h_s2t), h_t2s.stride(0), L.ptr(a_t2s), L.ptr(a_s2t), L.ptr(csr.rowptr), L.ptr(csr.col),
            L.ptr(mask_u8), L.ptr(t_rowptr), L.ptr(t_eid), L.ptr(t_dst), csr.num_nodes, csr.num_edges, D, float(negative_slope),
            L.ptr(out), out.stride(0), L.ptr(alpha), L.ptr(grad_out), grad_out.stride(0),
            L.ptr(dh_t2s), L.ptr(dh_s2t), L.ptr(da_t2s), L.ptr(da_s2t), L.ptr(ws), wsb, L.stream())
        L.check(rc, "bgnn_adaptedconv_aggregate_bwd_pull_f32")
        return dh_t2s, dh_s2t, da_t2s, da_s2t
    dh_t2s, dh_s2t = torch.zeros_like(h_t2s), torch.zeros_like(h_s2t)
    rc = lib.bgnn_adaptedconv_aggregate_bwd_f32(
        L.ptr(h_t2s), L.ptr(h_s2t), h_t2s.stride(0), L.ptr(a_t2s), L.ptr(a_s2t), L.ptr(csr.rowptr), L.ptr(csr.col),
        L.ptr(mask_u8), 0, csr.num_nodes, D, float(negative_slope), L.ptr(out), out.stride(0), L.ptr(alpha),
        L.ptr(grad_out), grad_out.stride(0), L.ptr(dh_t2s), L.ptr(dh_s2t), L.ptr(da_t2s), L.ptr(da_s2t), L.stream())
    L.check(rc, "bgnn_adaptedconv_aggregate_bwd_f32")
    return dh_t2s, dh_s2t, da_t2s, da_s2t


def adaptedconv_aggregate_heads_bwd(t2s, s2t, a_t2s, a_s2t, csr, mask_u8, D, heads, out, state_ms, grad_out, log_softmax,
                                    negative_slope=0.1):
    """backward of `adaptedconv_aggregate(..., heads=2|3, part=3)` for interleaved narrow heads ([N, heads*4] tables):
    -> (dh_t2s, dh_s2t [N, heads*4], da_t2s, da_s2t [heads, D]); one CSR walk per pass for all heads."""
    lib = L.lib()
    dev = t2s.device
    N = csr.num_nodes
    assert t2s.shape == (N, heads * 4) and t2s.is_contiguous() and s2t.is_contiguous() and out.is_contiguous() and grad_out.is_contiguous()
    da_t2s = torch.zeros(heads, D, dtype=torch.float32, device=dev)
    da_s2t = torch.zeros(heads, D, dtype=torch.float32, device=dev)
    dh_t2s, dh_s2t = torch.empty_like(t2s), torch.empty_like(s2t)
    t_rowptr, t_eid, t_dst = csr.transposed()
    dh_, sh_ = (csr.hub_tables(), csr.transposed_hub_tables()) if os.environ.get("BGNN_HUB_ROWS", "1") != "0" else (None, None)
    if dh_ is not None or sh_ is not None:          # graphs with hub rows: segments + merges (bgnn.h)
        none4 = (None, None, None, None)
        d_rows, d_ptr, d_bounds, d_node = dh_ if dh_ is not None else none4
        s_rows, s_ptr, s_bounds, s_node = sh_ if sh_ is not None else none4
        nd = 0 if dh_ is None else int(d_node.numel())
        ns = 0 if sh_ is None else int(s_node.numel())
        wsb = lib.bgnn_aggregate_heads_bwd_hub_workspace_bytes(N, csr.num_edges, heads, nd, ns)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        rc = lib.bgnn_adaptedconv_aggregate_heads_bwd_hub_f32(
            L.ptr(t2s), L.ptr(s2t), L.ptr(a_t2s), L.ptr(a_s2t), L.ptr(csr.rowptr), L.ptr(csr.col), L.ptr(mask_u8),
            L.ptr(t_rowptr), L.ptr(t_dst), N, csr.num_edges, D, heads, float(negative_slope),
            L.ptr(out), L.ptr(state_ms), L.ptr(grad_out), int(bool(log_softmax)), L.ptr(dh_t2s), L.ptr(dh_s2t),
            L.ptr(da_t2s), L.ptr(da_s2t), HUB_THRESHOLD,
            L.ptr(d_rows), 0 if dh_ is None else int(d_rows.numel()), L.ptr(d_ptr), L.ptr(d_bounds), L.ptr(d_node), nd,
            L.ptr(s_rows), 0 if sh_ is None else int(s_rows.numel()), L.ptr(s_ptr), L.ptr(s_bounds), L.ptr(s_node), ns,
            L.ptr(ws), wsb, L.stream())
        L.check(rc, "bgnn_adaptedconv_aggregate_heads_bwd_hub_f32")
        return dh_t2s, dh_s2t, da_t2s, da_s2t
    wsb = lib.bgnn_aggregate_heads_bwd_workspace_bytes(N, csr.num_edges, heads)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    rc = lib.bgnn_adaptedconv_aggregate_heads_bwd_f32(
        L.ptr(t2s), L.ptr(s2t), L.ptr(a_t2s), L.ptr(a_s2t), L.ptr(csr.rowptr), L.ptr(csr.col), L.ptr(mask_u8),
        L.ptr(t_rowptr), L.ptr(t_eid), L.ptr(t_dst), N, csr.num_edges, D, heads, float(negative_slope),
        L.ptr(out), L.ptr(state_ms), L.ptr(grad_out), int(bool(log_softmax)), L.ptr(dh_t2s), L.ptr(dh_s2t),
        L.ptr(da_t2s), L.ptr(da_s2t), L.ptr(ws), wsb, L.stream())
    L.check(rc, "bgnn_adaptedconv_aggregate_heads_bwd_f32")
    return dh_t2s, dh_s2t, da_t2s, da_s2t


def l2_normalize_rows(q, eps=1e-8):
    out = torch.empty_like(q)
    rc = L.lib().bgnn_l2_normalize_rows_f32(L.ptr(q), q.shape[0], q.shape[1], float(eps), L.ptr(out), L.stream())
    L.check(rc, "bgnn_l2_normalize_rows_f32")
    return out


_COS_D = (32, 64, 128, 256)


def _pad_cols(t, d):
    if t.shape[1] == d:
        return t.contiguous()
    out = torch.zeros(t.shape[0], d, dtype=t.dtype, device=t.device)
    out[:, : t.shape[1]] = t
    return out


def cosine_topk(qn_query, qn_cand, k, apply_sigmoid=True):
    """Top-k cosine neighbours of every query among the candidates (both already L2-normalised).
    -> (idx int64 [Nq,k], val fp32 [Nq,k], n_fallback int32[2] = (rows re-done exhaustively, rows sent to the precise pass))."""
    lib = L.lib()
    d = qn_query.shape[1]
    dk = next((c for c in _COS_D if c >= d), None)
    if dk is None:
        raise ValueError("embedding width > 256 unsupported")
    qq, qc = _pad_cols(qn_query, dk), _pad_cols(qn_cand, dk)   # zero columns do not change a dot product
    Nq, Nc = qq.shape[0], qc.shape[0]
    dev = qq.device
    idx = torch.empty(Nq, k, dtype=torch.int64, device=dev)
    val = torch.empty(Nq, k, dtype=torch.float32, device=dev)
    nfb = torch.zeros(2, dtype=torch.int32, device=dev)
    wsb = lib.bgnn_topk_workspace_bytes(Nq, Nc, k)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    rc = lib.bgnn_cosine_topk_f32(L.ptr(qq), L.ptr(qc), Nq, Nc, dk, k, 1 if apply_sigmoid else 0, L.ptr(idx),
                                  L.ptr(val), L.ptr(nfb), L.ptr(ws), wsb, L.stream())
    L.check(rc, "bgnn_cosine_topk_f32")
    return idx, val, nfb


def mlp_pair_topk(A_cand, B_query, bn_scale, bn_shift, w2, b2, k, apply_sigmoid=True):
    lib = L.lib()
    Nq, Nc, H = B_query.shape[0], A_cand.shape[0], A_cand.shape[1]
    dev = A_cand.device
    idx = torch.empty(Nq, k, dtype=torch.int64, device=dev)
    val = torch.empty(Nq, k, dtype=torch.float32, device=dev)
    nfb = torch.zeros(2, dtype=torch.int32, device=dev)
    wsb = lib.bgnn_topk_workspace_bytes(Nq, Nc, k)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    rc = lib.bgnn_mlp_pair_topk_f32(L.ptr(A_cand), L.ptr(B_query), L.ptr(bn_scale), L.ptr(bn_shift), L.ptr(w2),
                                    float(b2), Nq, Nc, H, k, 1 if apply_sigmoid else 0, L.ptr(idx), L.ptr(val),
                                    L.ptr(nfb), L.ptr(ws), wsb, L.stream())
    L.check(rc, "bgnn_mlp_pair_topk_f32")
    return idx, val, nfb


def topk_edges(idx, cand_base=0, query_base=0):
    """[Nq,k] top-k table -> edge_index [2, Nq*k] with (from = candidate, to = query)."""
    Nq, k = idx.shape
    out = torch.empty(2, Nq * k, dtype=torch.int64, device=idx.device)
    rc = L.lib().bgnn_topk_edges_i64(L.ptr(idx), Nq, k, int(cand_base), int(query_base), L.ptr(out), L.stream())
    L.check(rc, "bgnn_topk_edges_i64")
    return out


def topk_edges_coalesced(idx, n_cand, cand_base=0, query_base=0):
    """coalesce(topk_edges(idx)) in one call for a table of DISTINCT valid candidates per query (0 <= idx < n_cand,
    k <= n_cand: what the top-k kernels return): a stable 32-bit pair sort by candidate id, no device-to-host read."""
    Nq, k = idx.shape
    lib = L.lib()
    out = torch.empty(2, Nq * k, dtype=torch.int64, device=idx.device)
    wsb = lib.bgnn_topk_edges_coalesced_workspace_bytes(Nq, k)
    ws = torch.empty(wsb, dtype=torch.uint8, device=idx.device)
    rc = lib.bgnn_topk_edges_coalesced_i64(L.ptr(idx), Nq, k, int(n_cand), int(cand_base), int(query_base), L.ptr(out), L.ptr(ws), wsb,
                                           L.stream())
    L.check(rc, "bgnn_topk_edges_coalesced_i64")
    return out


def gather_rows(src, idx, out=None):
    """out[r] = src[idx[r]] for a 2-D float32 table with unit column stride and a row length that is a multiple of 4
    (the halo send lists; `index_select` needs 44 us for 2e5 rows of 48 bytes, this kernel the time of the bytes)."""
    n, w = int(idx.shape[0]), int(src.shape[1])
    if out is None:
        out = torch.empty(n, w, dtype=torch.float32, device=src.device)
    assert src.dtype == torch.float32 and idx.dtype == torch.int64 and w % 4 == 0 and out.shape == (n, w)
    rc = L.lib().bgnn_gather_rows_f32(L.ptr_rows(src), src.shape[0], src.stride(0), L.ptr(idx), n, w, L.ptr_rows(out),
                                      out.stride(0), L.stream())
    L.check(rc, "bgnn_gather_rows_f32")
    return out


def coalesce(edge_index, num_nodes=None):
    """torch_geometric.utils.coalesce on the GPU (sorted by row*n+col, duplicates dropped)."""
    lib = L.lib()
    ei = edge_index.contiguous().clone()
    E = int(ei.shape[1])
    if E == 0:
        return ei
    n = int(num_nodes) if num_nodes is not None else int(ei.max().item()) + 1
    dev = ei.device
    e_out = torch.zeros(1, dtype=torch.int64, device=dev)
    wsb = lib.bgnn_coalesce_workspace_bytes(E)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    rc = lib.bgnn_coalesce_i64(L.ptr(ei), E, n, L.ptr(e_out), L.ptr(ws), wsb, L.stream())
    L.check(rc, "bgnn_coalesce_i64")
    ne = int(e_out.item())
    return ei[:, :ne].contiguous()
