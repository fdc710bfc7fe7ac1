"""GPU: round-3 kernels -- the 32-bit-addressed plain wide aggregation (agg_wide_fast_kernel, ABI 113) against the general
kernel bit for bit and against the fp64 oracle."""
import numpy as np
import pytest
import torch

from conftest import assert_close
from oracle import oracle_c as OC
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _general(tS, tT, a1, a2, csr, m8, D, slope, sc, sh, relu, sums, row_begin, row_end, out, alpha=None):
    """bgnn_adaptedconv_aggregate_f32: no table_rows promise -> always the general kernel (agg_wide_kernel for D > 32)"""
    from bridged_gnn_amd import _lib as L, ops
    tq = ops._tile_queue(tS.device)
    rc = L.lib().bgnn_adaptedconv_aggregate_f32(
        L.ptr_rows(tS), L.ptr_rows(tT), tS.stride(0), L.ptr(a1), L.ptr(a2), L.ptr(csr.rowptr), L.ptr(csr.col), L.ptr(m8),
        row_begin, row_end, D, slope, L.ptr_rows(out), out.stride(0), L.ptr(alpha), L.ptr(sc), L.ptr(sh), 1 if relu else 0, None, 0, row_begin, 1,
        L.ptr(sums), L.ptr(tq), L.stream())
    L.check(rc, "bgnn_adaptedconv_aggregate_f32")
    return out


@pytest.mark.parametrize("D,n,deg,seed", [(128, 5000, 9, 0), (100, 3001, 5, 1), (36, 777, 3, 2), (64, 4096, 12, 3), (200, 1500, 7, 4),
                                          (256, 900, 4, 5), (128, 9, 2, 6), (128, 20000, 21, 7)])
def test_fast_wide_aggregation_is_bit_identical_to_the_general_kernel(D, n, deg, seed):
    """Same arithmetic, different addressing: the two kernels must agree in every bit of the rows (the column sums are order-dependent
    fp32 partials in both); both meet the 1e-5 bar against the fp64 evaluation of KTGNN.py:292-305."""
    from bridged_gnn_amd import ops, synth
    rng = np.random.default_rng(3000 + seed)
    ei, mask = synth.random_multigraph(n, deg * n, frac_src=float(rng.uniform(0.2, 0.8)), n_isolated=min(3, n // 4), seed=seed)
    slope = float(rng.choice([0.0, 0.1, 0.2, 1.0]))
    hS, hT = rng.standard_normal((n, D)).astype(np.float32), rng.standard_normal((n, D)).astype(np.float32)
    a1, a2 = (rng.standard_normal(D) * 0.3).astype(np.float32), (rng.standard_normal(D) * 0.3).astype(np.float32)
    csr = ops.build_dst_csr(_t(ei), n)
    assert csr.hub_tables() is None                      # (hub rows take the general kernel: test_degree_skew_hub_rows)
    ld = ops.pad4(D)
    both = torch.zeros(2, n, ld, device=DEV)             # one allocation: the two tables share a window
    tS, tT = both[0], both[1]
    tS[:, :D], tT[:, :D] = _t(hS), _t(hT)
    m8 = _t(mask).to(torch.uint8)
    sc = _t(rng.uniform(0.5, 1.5, D).astype(np.float32)); sh = _t(rng.standard_normal(D).astype(np.float32))
    a1t, a2t = _t(a1), _t(a2)
    for relu, with_ep, (rb, re) in ((True, True, (0, n)), (False, False, (0, n)), (True, True, (n // 3, n - n // 5))):
        s_fast = torch.zeros(2 * ld + 2, dtype=torch.float64, device=DEV)
        s_gen = torch.zeros_like(s_fast)
        o_fast = torch.full((n, ld), 7.0, device=DEV)
        o_gen = torch.full((n, ld), 7.0, device=DEV)
        ep = dict(ep_scale=sc, ep_shift=sh) if with_ep else {}
        ops.adaptedconv_aggregate(tS, tT, a1t, a2t, csr, m8, D, slope, ep_relu=relu, colsum=s_fast, out=o_fast, row_begin=rb, row_end=re, **ep)
        _general(tS, tT, a1t, a2t, csr, m8, D, slope, sc if with_ep else None, sh if with_ep else None, relu, s_gen, rb, re, o_gen)
        torch.cuda.synchronize()
        assert torch.equal(o_fast, o_gen), f"D={D} n={n} rows [{rb},{re}) relu={relu}: {(o_fast != o_gen).sum().item()} elements differ"
        assert (o_fast[:rb] == 7.0).all() and (o_fast[re:] == 7.0).all()             # rows outside the range untouched
        # (column sums: fp32 partials per lane over the rows the dynamic tile queue handed it -- order-dependent in both kernels)
        torch.testing.assert_close(s_fast, s_gen, rtol=2e-6, atol=1e-3)
        assert s_fast[2 * ld].item() == mask[rb:re].sum() and s_fast[2 * ld + 1].item() == (~mask[rb:re]).sum()
    # the training forward's launch: attention coefficients in CSR order, written by the same lanes in the same order
    o_fast, al_fast = ops.adaptedconv_aggregate(tS, tT, a1t, a2t, csr, m8, D, slope, want_alpha=True)
    al_gen = torch.full_like(al_fast, -1.0)
    o_gen = _general(tS, tT, a1t, a2t, csr, m8, D, slope, None, None, False, None, 0, n, torch.empty_like(o_fast), alpha=al_gen)
    assert torch.equal(o_fast, o_gen) and torch.equal(al_fast, al_gen)
    rp = csr.rowptr.long()
    seg = torch.repeat_interleave(torch.arange(n, device=DEV), rp[1:] - rp[:-1])
    torch.testing.assert_close(torch.zeros(n, device=DEV).index_add_(0, seg, al_fast), torch.ones(n, device=DEV), rtol=0, atol=2e-6)
    rowptr, col, _ = O.dst_csr(ei, mask)
    truth = OC.adaptedconv_aggregate_f64(hS, hT, a1, a2, rowptr, col, mask, slope=slope)
    out = ops.adaptedconv_aggregate(tS, tT, a1t, a2t, csr, m8, D, slope)
    assert_close(out[:, :D].cpu().numpy(), truth, what=f"fast wide aggregation vs fp64 truth D={D} n={n}")
    assert (out[:, D:] == 0).all()


def test_fast_wide_aggregation_with_a_window_above_2gb():
    """Tables 2.5 GB apart: the window's byte count no longer fits a signed 32-bit integer (the buffer descriptor's num_records is unsigned) --
    still the fast kernel, still the same bits as with adjacent tables."""
    from bridged_gnn_amd import ops, synth
    n, D = 3000, 128
    rng = np.random.default_rng(6)
    ei, mask = synth.random_multigraph(n, 7 * n, frac_src=0.4, n_isolated=2, seed=12)
    csr = ops.build_dst_csr(_t(ei), n)
    gap = (5 << 29) // 4                                   # 2.5 GB in floats
    big = torch.empty(gap + n * D, dtype=torch.float32, device=DEV)
    tS, tT = big[:n * D].view(n, D), big[gap:].view(n, D)
    tS.copy_(_t(rng.standard_normal((n, D)).astype(np.float32))); tT.copy_(_t(rng.standard_normal((n, D)).astype(np.float32)))
    a1, a2 = _t((rng.standard_normal(D) * 0.3).astype(np.float32)), _t((rng.standard_normal(D) * 0.3).astype(np.float32))
    m8 = _t(mask).to(torch.uint8)
    assert (1 << 31) < tT.data_ptr() - tS.data_ptr() + n * D * 4 < (1 << 32)
    wide = ops.adaptedconv_aggregate(tS, tT, a1, a2, csr, m8, D, 0.1)
    swapped = ops.adaptedconv_aggregate(tT, tS, a2, a1, csr, (1 - m8).contiguous(), D, 0.1)      # the other table on top: same function of the graph with domains swapped
    both = torch.stack([tS, tT])
    near = ops.adaptedconv_aggregate(both[0], both[1], a1, a2, csr, m8, D, 0.1)
    assert torch.equal(wide, near) and torch.equal(swapped, near)


def test_fast_wide_aggregation_falls_back_when_the_tables_are_4gb_apart():
    """Two tables whose window exceeds 32 bits of byte offsets take the general kernel: same result, no fault."""
    from bridged_gnn_amd import ops, synth
    n, D = 2000, 128
    rng = np.random.default_rng(5)
    ei, mask = synth.random_multigraph(n, 6 * n, frac_src=0.5, n_isolated=2, seed=11)
    csr = ops.build_dst_csr(_t(ei), n)
    big = torch.empty((1 << 30) + (1 << 18) + n * D, dtype=torch.float32, device=DEV)   # the tables 4 GB + 1 MB apart, one allocation
    tS, tT = big[:n * D].view(n, D), big[(1 << 30) + (1 << 18):].view(n, D)
    tS.copy_(_t(rng.standard_normal((n, D)).astype(np.float32))); tT.copy_(_t(rng.standard_normal((n, D)).astype(np.float32)))
    a1, a2 = _t((rng.standard_normal(D) * 0.3).astype(np.float32)), _t((rng.standard_normal(D) * 0.3).astype(np.float32))
    m8 = _t(mask).to(torch.uint8)
    assert tT.data_ptr() - tS.data_ptr() > (1 << 32)
    far = ops.adaptedconv_aggregate(tS, tT, a1, a2, csr, m8, D, 0.1)
    both = torch.stack([tS, tT])
    near = ops.adaptedconv_aggregate(both[0], both[1], a1, a2, csr, m8, D, 0.1)
    assert torch.equal(far, near)


@pytest.mark.parametrize("D,n,deg,slope,seed", [(128, 20000, 21, 0.1, 0), (100, 3001, 5, 0.2, 1), (68, 777, 3, 0.0, 2), (128, 9, 2, 1.0, 3),
                                                (96, 4096, 12, 0.1, 4), (128, 1500, 30, 0.1, 5)])
def test_fast_pull_backward_equals_the_atomic_backward(D, n, deg, slope, seed):
    """agg_bwd_dst_fast_kernel / agg_bwd_src_fast_kernel (64 < D <= 128, no hub rows: what the training step's hidden conv issues)
    against the atomic scatter form of the same backward (its own parity vs autograd: tests/test_gpu_training.py), incl. isolated
    rows, a row count that is no multiple of the tile, pad columns (D % 4 != 0 is covered by 100 / 68 -> ld 100 / 68) and slope 0 / 1."""
    from bridged_gnn_amd import _lib, ops, synth
    rng = np.random.default_rng(4000 + seed)
    ei, mask = synth.random_multigraph(n, deg * n, frac_src=float(rng.uniform(0.2, 0.8)), n_isolated=min(3, n // 4), seed=seed)
    csr = ops.build_dst_csr(_t(ei), n)
    assert csr.hub_tables() is None and csr.transposed_hub_tables() is None
    ld = ops.pad4(D)
    both = torch.zeros(2, n, ld, device=DEV)
    hS, hT = both[0], both[1]
    hS[:, :D], hT[:, :D] = _t(rng.standard_normal((n, D)).astype(np.float32)), _t(rng.standard_normal((n, D)).astype(np.float32))
    a1, a2 = _t((rng.standard_normal(D) * 0.3).astype(np.float32)), _t((rng.standard_normal(D) * 0.3).astype(np.float32))
    m8 = _t(mask).to(torch.uint8)
    out, alpha = ops.adaptedconv_aggregate(hS, hT, a1, a2, csr, m8, D, slope, want_alpha=True)
    g = torch.zeros(n, ld, device=DEV); g[:, :D] = _t(rng.standard_normal((n, D)).astype(np.float32))
    pull = ops.adaptedconv_aggregate_bwd(hS, hT, a1, a2, csr, m8, D, out, alpha, g, slope)
    again = ops.adaptedconv_aggregate_bwd(hS, hT, a1, a2, csr, m8, D, out, alpha, g, slope)
    assert torch.equal(pull[0], again[0]) and torch.equal(pull[1], again[1]), "every dH row is written once: deterministic"
    L = _lib.lib()
    d1, d2 = torch.zeros_like(hS), torch.zeros_like(hT)
    da1, da2 = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    rc = L.bgnn_adaptedconv_aggregate_bwd_f32(_lib.ptr_rows(hS), _lib.ptr_rows(hT), ld, _lib.ptr(a1), _lib.ptr(a2), _lib.ptr(csr.rowptr), _lib.ptr(csr.col),
                                              _lib.ptr(m8), 0, n, D, slope, _lib.ptr(out), out.stride(0), _lib.ptr(alpha), _lib.ptr(g), ld,
                                              _lib.ptr(d1), _lib.ptr(d2), _lib.ptr(da1), _lib.ptr(da2), _lib.stream())
    assert rc == 0
    rel = lambda a, b: ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()
    for name, a, b in (("dh_t2s", pull[0], d1), ("dh_s2t", pull[1], d2), ("da_t2s", pull[2], da1), ("da_s2t", pull[3], da2)):
        assert rel(a, b) < 2e-5, (name, D, n, rel(a, b))           # (the atomic sums are order-dependent in the last bits: the bar of the older pair test)
    assert (pull[0][:, D:] == 0).all() and (pull[1][:, D:] == 0).all()


@pytest.mark.parametrize("n,feat,classes,s_only_bridges", [(3, 128, 2, False), (33, 128, 2, True), (257, 100, 3, True), (1025, 128, 4, False),
                                                           (4099, 72, 2, True)])
def test_round3_forward_kernels_at_edge_sizes_vs_c_oracle(n, feat, classes, s_only_bridges):
    """The eval forward at hidden 128 -- i.e. through transform_stream_kernel (tile_need on when the graph has s -> t bridges only),
    agg_wide_fast_kernel, cls_stage_kernel and the three-head walk -- on graphs of 3 ... 4099 nodes (fewer rows than one tile, one row
    past a tile, fewer tiles than CUs), input widths 72 / 100 / 128, 2-4 classes, against the C oracle's full forward at the default bar."""
    from bridged_gnn_amd import synth
    from bridged_gnn_amd.data import Data
    from bridged_gnn_amd.ktgnn import KTGNN_no_complement
    rng = np.random.default_rng(n)
    ns = max(n // 2, 1)
    if s_only_bridges:                                     # [sources ; targets], within-domain edges + source -> target bridges only
        src_w = rng.integers(0, ns, 3 * ns); dst_w = rng.integers(0, ns, 3 * ns)
        tw_s = ns + rng.integers(0, n - ns, 3 * (n - ns)); tw_d = ns + rng.integers(0, n - ns, 3 * (n - ns))
        br_s = rng.integers(0, ns, 5 * (n - ns)); br_d = ns + rng.integers(0, n - ns, 5 * (n - ns))
        ei = np.stack([np.concatenate([src_w, tw_s, br_s]), np.concatenate([dst_w, tw_d, br_d])]).astype(np.int64)
        mask = np.arange(n) < ns
    else:
        ei, mask = synth.random_multigraph(n, 6 * n, frac_src=0.5, n_isolated=min(2, n - 1), seed=n)
    x = rng.standard_normal((n, feat)).astype(np.float32)
    torch.manual_seed(n)
    model = KTGNN_no_complement(feat, classes, 2, 128, root_weight=False, use_bn=True, dim_share=feat, need_complement=False)
    g = torch.Generator().manual_seed(3)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    model = model.to(DEV).eval()
    data = Data(x=_t(x), edge_index=_t(ei), central_mask=_t(mask))
    with torch.no_grad():
        out = [t.cpu().numpy() for t in model(data)[:3]]
        emb = model.get_emb(data).cpu().numpy()[:, :128]
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    rowptr, col, _ = O.dst_csr(ei, mask)
    rb, rt, rth, remb = OC.ktgnn_forward_eval(x, rowptr, col, mask, sd, return_emb=True)
    for name, got, ref in (("hidden conv (BN+ReLU)", emb, remb), ("logp_base", out[0], rb), ("logp_target", out[1], rt), ("logp_target_hat", out[2], rth)):
        assert_close(got, ref, what=f"{name} n={n} feat={feat}")


@pytest.mark.parametrize("din,dout,n,relu", [(8, 128, 4097, False), (4, 64, 333, True), (12, 128, 1000, False), (16, 256, 777, True), (8, 128, 1, False)])
def test_linear_with_a_skinny_reduction(din, dout, n, relu):
    """bgnn_linear_f32 at Din <= 16 (the input gradient Gall . Wcat of a narrow conv's transform backward, K = 8): against fp64 at the default
    bar, row-strided input view included.  (A VALU kernel for this shape measured 0.20 ms against the W-stationary MFMA kernel's 0.13: not kept.)"""
    from bridged_gnn_amd import ops
    rng = np.random.default_rng(din * 1000 + dout + n)
    xb = _t(rng.standard_normal((n, din + 4)).astype(np.float32))
    x = xb[:, :din]                                                   # row stride din + 4
    W = _t((rng.standard_normal((dout, din)) * 0.3).astype(np.float32))
    b = _t(rng.standard_normal(dout).astype(np.float32))
    got = ops.linear(x, W, b, relu=relu).cpu().numpy()
    ref = x.double().cpu().numpy() @ W.double().cpu().numpy().T + b.double().cpu().numpy()
    ref = np.maximum(ref, 0) if relu else ref
    assert_close(got, ref, what=f"linear din={din} dout={dout} n={n}")
