"""Partitioned TRAINING step of KT-GNN (reference loop main_graph_knowledge_transfer.py:39-68, 300 epochs of one full-batch step)
on a destination-node partition; new -- the reference is single-device.

Every rank owns a block of destination rows and all their in-edges (`dist.PartitionPlan`).  The training-mode forward of
`KTGNN_no_complement` (KTGNN.py:401-435) runs on the rank's rows with the single-GPU autograd functions of `ktgnn.py` (fused HIP
transform / aggregation forward and backward) over an "extended" local graph: the rank's rows followed by its halo rows, made
square by giving the halo rows no in-edges.  What crosses ranks, forward and backward:
  * conv 0 reads the graph's INPUT features: their halo rows are resident (fetched once, `PartitionedKTGNN._input_ext`), the rank
    transforms local + halo rows itself -- no exchange; the halo rows' share of the weight gradient is part of this rank's loss;
  * later convs and the classifier stage transform the rank's own rows and send the transformed rows their consumers reference
    through ONE all_to_all (48-byte rows for the three classifier convs); its autograd backward is the REVERSE exchange of the
    gradient rows (dH after the aggregation's pass B), added into the owners' rows by autograd;
  * the per-domain sums behind delta (KTGNN.py:275) are all-reduced; the gradient through the domain means is a global quantity:
    every transform hands its local adjoint of delta to a hook that all-reduces it and applies +1/n_S | -1/n_T on the owned rows;
  * train-mode BatchNorm uses the statistics of ALL N nodes (KTGNN.py:420-430): sum / sum-of-squares all-reduced forward, the two
    backward reductions all-reduced too (`_SyncBnReluDrop`);
  * parameter gradients are summed over ranks in ONE bucketed all-reduce (`sync_grads`).
Losses are sums over owned rows with GLOBAL normalisers (`reference_loss`), so the per-rank losses add up to the reference's loss.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import ops
from .dist import PartitionPlan
from .ktgnn import _AggregateFn, _AggregateHeadsFn, _pad_cols4


class _Comm:
    """collectives on device tensors; a gloo group with CUDA tensors (ranks rehearsing on one GPU) stages through the host"""

    def __init__(self, group, device, world):
        self.group, self.world = group, world
        self.live = dist.is_initialized() and world > 1
        self.host = self.live and torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo"

    def all_reduce(self, t):
        if not self.live:
            return t
        if self.host:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            return h.to(t.device)
        dist.all_reduce(t, group=self.group)
        return t

    def all_to_all(self, send, send_splits, recv_splits):
        n = int(sum(recv_splits))
        recv = torch.empty((n,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        if not self.live:
            return recv
        if self.host:
            r = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(r, send.contiguous().cpu(), output_split_sizes=list(recv_splits), input_split_sizes=list(send_splits),
                                   group=self.group)
            return r.to(send.device)
        dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=list(recv_splits), input_split_sizes=list(send_splits),
                               group=self.group)
        return recv


class _HaloRows(torch.autograd.Function):
    """rows (in send order) -> halo rows (in halo order); backward: the reverse exchange of the gradient rows"""

    @staticmethod
    def forward(ctx, send, comm, send_splits, recv_splits):
        ctx.cfg = (comm, send_splits, recv_splits)
        return comm.all_to_all(send, send_splits, recv_splits)

    @staticmethod
    def backward(ctx, g):
        comm, send_splits, recv_splits = ctx.cfg
        return comm.all_to_all(g.contiguous(), recv_splits, send_splits), None, None, None


class _SyncBnReluDrop(torch.autograd.Function):
    """BatchNorm1d(train) -> ReLU -> dropout over the rows of ALL ranks (KTGNN.py:420-430; clf_transformer's BN + ReLU, :407-411)"""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, relu, p_drop, comm, n_global):
        D = x.shape[1]
        xd = x.double()
        st = comm.all_reduce(torch.cat((xd.sum(0), (xd * xd).sum(0))))
        mean = st[:D] / n_global
        var = (st[D:] / n_global - mean * mean).clamp_min(0)                    # biased, as BatchNorm normalises
        rstd = torch.rsqrt(var + bn.eps)
        xhat = ((xd - mean) * rstd).float()
        y = xhat * weight + bias if weight is not None else xhat
        keep = torch.ones_like(y)
        if relu:
            keep = keep * (y > 0)
        if p_drop > 0:
            keep = keep * (torch.rand_like(y) >= p_drop) / (1.0 - p_drop)
        if bn.track_running_stats and bn.running_mean is not None:
            with torch.no_grad():
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                bn.num_batches_tracked.add_(1)
                bn.running_mean.mul_(1 - mom).add_(mean.float(), alpha=mom)
                bn.running_var.mul_(1 - mom).add_((var * n_global / max(n_global - 1, 1)).float(), alpha=mom)
                bn._bgnn_affine = None
        ctx.save_for_backward(xhat, rstd.float(), keep, weight)
        ctx.cfg = (comm, n_global)
        return y * keep

    @staticmethod
    def backward(ctx, gy):
        xhat, rstd, keep, weight = ctx.saved_tensors
        comm, n = ctx.cfg
        g = gy * keep
        dxh = g * weight if weight is not None else g
        D = g.shape[1]
        red = comm.all_reduce(torch.cat((dxh.double().sum(0), (dxh * xhat).double().sum(0)))).float()
        dx = rstd * (dxh - red[:D] / n - xhat * (red[D:] / n))
        dw = (g * xhat).sum(0) if weight is not None else None
        db = g.sum(0) if weight is not None else None
        return dx, dw, db, None, None, None, None, None


class PartitionedTrainer:
    """training-mode forward / backward of `model` (a `KTGNN_no_complement`) on rank `rank`'s rows; see the module docstring.

        tr = PartitionedTrainer(model, edge_index, central_mask, rank, world, device)
        out = tr.forward(x[tr.owned_global])            # (logp_base, logp_target, logp_target_hat) of the owned rows
        loss = tr.reference_loss(out, y[tr.owned_global], train_mask[tr.owned_global])
        opt.zero_grad(); loss.backward(); tr.sync_grads(); opt.step()
    """

    def __init__(self, model, edge_index, central_mask, rank, world, device, owner=None, group=None):
        self.model, self.rank, self.world, self.device, self.group = model, rank, world, device, group
        p = self.plan = PartitionPlan(edge_index, central_mask, rank, world, owner=owner)
        self.comm = _Comm(group, device, world)
        t = lambda a: torch.from_numpy(a).to(device)
        self.owned_global = t(p.owned_global)
        self.n_local, self.n_halo, self.n_ext = p.n_local, p.n_halo, p.n_local + p.n_halo
        # square extended graph: the halo rows have no in-edges (their outputs are never used)
        rowptr = torch.cat((t(p.rowptr), torch.full((p.n_halo,), int(p.rowptr[-1]), dtype=torch.int32, device=device)))
        self.csr = ops.DstCSR(rowptr.contiguous(), t(p.col_ext), None, p.local_num_edges, self.n_ext)
        self.send_rows_local, self.halo_ext_perm = t(p.send_rows_local), t(p.halo_ext_perm)
        self.mask_local = t(p.mask_local)
        self.mask_u8 = self.mask_local.to(torch.uint8).contiguous()
        self.mask_ext_u8 = torch.cat((self.mask_local, t(p.halo_mask[p.halo_ext_perm]))).to(torch.uint8).contiguous()
        m = torch.as_tensor(central_mask).bool()
        self.n_global, self.n_s, self.n_t = int(m.shape[0]), int(m.sum()), int((~m).sum())
        self.coef = torch.where(self.mask_local, 1.0 / max(self.n_s, 1), -1.0 / max(self.n_t, 1)).float()[:, None]
        self._x_ext = self._x_sums = None

    # ---- pieces ------------------------------------------------------------------------------------------------------
    def _halo_of(self, rows_local):
        """differentiable: the halo rows (extended order) of a rank-local row tensor"""
        p = self.plan
        halo = _HaloRows.apply(rows_local.index_select(0, self.send_rows_local), self.comm, p.send_splits, p.recv_splits)
        return halo.index_select(0, self.halo_ext_perm)

    def _global_sums(self, x_local):
        """all-reduced per-domain column sums (+ node counts) of the owned rows: constants of the transform (their gradient
        travels through `_mean_hook`)"""
        return self.comm.all_reduce(ops.domain_sums(_pad_cols4(x_local.detach()), self.mask_u8))

    def _mean_hook(self, ddl):
        return self.coef * self.comm.all_reduce(ddl.contiguous())[None, :]

    def _bn(self, x, bn, relu, p_drop):
        return _SyncBnReluDrop.apply(x, bn.weight, bn.bias, bn, relu, float(p_drop), self.comm, self.n_global)

    def _input_ext(self, x_local):
        if self._x_ext is None or self._x_ext[0] is not x_local or self._x_ext[1] != x_local._version:
            xp = _pad_cols4(x_local.detach().float())
            with torch.no_grad():
                ext = torch.cat((xp, self._halo_of(xp)))
            self._x_ext = (x_local, x_local._version, ext, self._global_sums(xp))
        return self._x_ext[2], self._x_ext[3]

    # ---- forward -----------------------------------------------------------------------------------------------------
    def forward(self, x_local):
        m = self.model
        if not m.training:
            raise RuntimeError("PartitionedTrainer.forward is the TRAINING forward; use dist.PartitionedKTGNN for evaluation")
        C = m.clf_base.out_channels
        if not ops.heads_log_softmax_supported(3, C) or m.clf_base.root_weight or m.clf_base.normalize:
            raise NotImplementedError("partitioned training: classifier convs with <= 4 classes, root_weight / normalize off")
        nl = self.n_local
        h = None
        for ind, conv in enumerate(m.convs):
            if ind == 0:
                x_ext, sums = self._input_ext(x_local)
                t2s, s2t = conv._transform_autograd(x_ext, self.mask_ext_u8, sums)
            else:
                t2s_l, s2t_l = conv._transform_autograd(h, self.mask_u8, self._global_sums(h), self._mean_hook)
                both = torch.cat((t2s_l, s2t_l), dim=1)
                ld = t2s_l.shape[1]
                ext = torch.cat((both, self._halo_of(both)))
                t2s, s2t = ext[:, :ld].contiguous(), ext[:, ld:].contiguous()
            out = _AggregateFn.apply(t2s, s2t, conv.a_f_t2s.weight.reshape(-1), conv.a_f_s2t.weight.reshape(-1), self.csr,
                                     self.mask_ext_u8, conv.out_channels, conv.negative_slope)[:nl, :conv.out_channels]
            if m.use_bn:
                h = self._bn(out, m.bns[ind], True, m.dropout)
            else:
                h = F.dropout(F.relu(out), p=m.dropout, training=True)
            h = h.contiguous()
        # classifier stage (KTGNN.py:432-435): three narrow convs share the graph, one exchange of 96-byte rows
        sums_h = self._global_sums(h)
        l0, bn, _, l3 = m.clf_transformer
        xt = l3(self._bn(l0(h), bn, True, 0.0)).contiguous()
        sums_t = self._global_sums(xt)
        tabs = [m.clf_base._transform_autograd(h, self.mask_u8, sums_h, self._mean_hook),
                m.clf_target._transform_autograd(h, self.mask_u8, sums_h, self._mean_hook),
                m.clf_target._transform_autograd(xt, self.mask_u8, sums_t, self._mean_hook)]
        ld = tabs[0][0].shape[1]
        both = torch.cat([t[0] for t in tabs] + [t[1] for t in tabs], dim=1)      # [n_local, 6 ld]: t2s x3 | s2t x3
        ext = torch.cat((both, self._halo_of(both)))
        pairs = []
        for j in range(3):
            pairs += [ext[:, j * ld:(j + 1) * ld], ext[:, (3 + j) * ld:(4 + j) * ld]]
        cs = (m.clf_base, m.clf_target, m.clf_target)
        a_t = torch.stack([c.a_f_t2s.weight.reshape(-1) for c in cs])
        a_s = torch.stack([c.a_f_s2t.weight.reshape(-1) for c in cs])
        logp = _AggregateHeadsFn.apply(self.csr, self.mask_ext_u8, C, m.clf_base.negative_slope, a_t, a_s, *pairs)[:nl, :, :C]
        return logp[:, 0], logp[:, 1], logp[:, 2]

    # ---- loss / gradients ------------------------------------------------------------------------------------------------
    def reference_loss(self, out, y_local, train_mask_local):
        """this rank's share of the reference's loss (main_graph_knowledge_transfer.py:44-54): the three masked NLL terms and the
        KL term as sums over the OWNED rows divided by the GLOBAL counts -- the shares of all ranks add up to the reference's value"""
        lb, lt, lth = out
        tm = train_mask_local.bool()
        tmt = tm & ~self.mask_local
        cnt = self.comm.all_reduce(torch.stack((tm.sum(), tmt.sum())).double()).float().clamp_min(1)
        yi = y_local.clamp_min(0)[:, None]

        def nll(logp, w):
            return -(logp.gather(1, yi).squeeze(1) * w).sum()
        w_b, w_t = tm.float() / cnt[0], tmt.float() / cnt[1]
        kl = (lt.exp() * (lt - lth)).sum() / self.n_global                       # F.kl_div(lth, lt, log_target=True, 'batchmean')
        return (2 * nll(lb, w_b) + nll(lt, w_t) + nll(lth, w_t)) / 4 + kl

    def sync_grads(self):
        """sum the parameter gradients over the ranks: ONE all-reduce of one flat bucket"""
        ps = [p for p in self.model.parameters() if p.grad is not None]
        if not ps or not self.comm.live:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        flat = self.comm.all_reduce(flat)
        o = 0
        for p in ps:
            n = p.grad.numel()
            p.grad.copy_(flat[o:o + n].view_as(p.grad))
            o += n
