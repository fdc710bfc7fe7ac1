#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: aggregated edges/s of the KT-GNN eval forward
(4 AdaptedConv calls: hidden conv + 3 classifier convs, reference models/KTGNN.py:401-435) plus
kNN-bridge pairs/s (config C5) as an extra field.  One JSON line on rank 0.

  python bench.py --gpus 1 --steps 20 --warmup 5                 # config C4: 1M nodes / 20M edges, hidden 128
  python bench.py --config c3                                    # config C3: Twitter_Graph stand-in, F=300, hidden 128
  python bench.py --config c2                                    # config C2: Sync-RD_intra 10k nodes, hidden 64
  python bench.py --config c5                                    # config C5 alone: 100k x 100k cosine kNN bridge
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one full-graph eval forward with inputs resident in HBM.  N>1: the SAME graph is
node-partitioned over the ranks (strong scaling): one small all-reduce (domain sums of the hidden
activations) and one all_to_all of the classifier stage's halo rows per forward over RCCL; the halo rows and
the all-reduced domain sums of the input features (static data, like the graph) stay resident per version of x.
After the eager measurement the same K steps are timed as replays of a HIP graph of the forward
(outputs checked against the eager ones); the faster execution is `value`, both are in the line.
`parity` = the timed forward's own outputs against the CPU oracle (all rows at N=1, part of the cpu_baseline leg).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
L2_GATHER_GBS = 17800.0        # same guide, "Indexed rows": 16.8-18.8 TB/s chip-wide for row gathers served from the XCD L2s
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (the guide's ~2.5 PF)
PMC_DIRS = (os.path.join("profiles", "r03"), os.path.join("profiles", "r02"))     # newest round first


def agg_bytes(E, N, D):
    """B_agg(D) = E'(4D+4) + N(8D+4) + 4 algorithmic bytes per AdaptedConv aggregation."""
    return E * (4 * D + 4) + N * (8 * D + 4) + 4


def pmc_traffic(args, world, graph):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE
    need separate profiler runs -- profiles/r02/README.md -- they cannot be read live here).  Only reported when this
    run's workload is the one those passes measured; the file is named in the line (`traffic_source`)."""
    for d in PMC_DIRS:
        rel = os.path.join(d, f"pmc_traffic_{args.config}_{graph}.json")
        try:
            t = json.load(open(os.path.join(ROOT, rel)))
            w = t["workload"]
            if world == 1 and (w["nodes"], w["edges"], w["hidden"], w["graph"]) == (args.nodes, args.edges, args.hidden, graph):
                return float(t["hbm_bytes_per_launch"]), rel
        except Exception:
            pass
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 20; 200 for the sub-millisecond configs c2 / c3, whose 20 steps "
                    "would be a 6 ms region -- shorter than the host transients seen right after another process released the GPU)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default: 5; 20 for c2 / c3)")
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c4",
                    help="BASELINE.json config: c4 (default) is the one the metric is quoted on; c5 = kNN bridge alone")
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--edges", type=int, default=20_000_000)
    ap.add_argument("--hidden", type=int, default=None)
    ap.add_argument("--feat", type=int, default=None)
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--graph", choices=["local", "uniform"], default="local",
                    help="c4: local = bridged/kNN-like locality (p_local=0.9, clusters of 1024); uniform = adversarial")
    ap.add_argument("--no-uniform", action="store_true", help="c4: skip the uniform-random graph's hidden aggregation (the conservative roofline figure)")
    ap.add_argument("--no-knn", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--knn-n", type=int, default=100_000)
    ap.add_argument("--train-steps", type=int, default=-1,
                    help="also time this many training steps (fwd+bwd+Adam, reference loss); default: 5 for config c4 on one GPU, else 0")
    ap.add_argument("--force-dist", action="store_true", help="run the partitioned (RCCL) code path even at world size 1")
    ap.add_argument("--no-partitioned-check", action="store_true",
                    help="N=1: skip timing the same forward through the partitioned driver at world size 1 (`partitioned_path_n1`)")
    ap.add_argument("--no-input-halo-cache", action="store_true",
                    help="N>1: exchange transformed rows for the first conv on every forward instead of keeping the halo rows "
                         "of the (static) input features resident and transforming them locally")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = rehearsal of the N>1 code path with several ranks sharing one GPU (payload staged through the host)")
    ap.add_argument("--graph-replay", action="store_true", help="(default behaviour; kept for old command lines)")
    ap.add_argument("--no-graph-replay", action="store_true",
                    help="skip the HIP-graph phase: by default the forward (collectives included) is captured once after the "
                         "eager measurement, checked against the eager outputs and timed for the same K steps; the faster "
                         "of the two is `value`, both are reported")
    ap.add_argument("--graph-timeout", type=float, default=60.0,
                    help="seconds the HIP-graph phase may take before the eager result is printed and the process exits non-zero")
    args = ap.parse_args()
    dflt = {"c2": (64, 64), "c3": (300, 128), "c4": (128, 128), "c5": (128, 128)}[args.config]
    args.feat = dflt[0] if args.feat is None else args.feat
    args.hidden = dflt[1] if args.hidden is None else args.hidden
    small = args.config in ("c2", "c3")
    args.steps = (200 if small else 20) if args.steps is None else args.steps
    args.warmup = (20 if small else 5) if args.warmup is None else args.warmup
    return args


# ------------------------------------------------------------------------------------------------ workloads
def c4_graph(nodes, edges, graph="local"):
    """C4: per node 6 within-domain in-neighbours, per target node 20 source in-neighbours, the rest random
    intra-domain edges; `local` draws a neighbour from the node's own cluster of 1024 consecutive ids w.p. 0.9."""
    from bridged_gnn_amd import synth
    n_src = nodes // 2
    n_tar = nodes - n_src
    extra = edges - 6 * nodes - 20 * n_tar
    return synth.bridged_graph(n_src, n_tar, k_within=6, k_cross=20, n_extra=max(extra, 0), cluster=1024,
                               p_local=0.9 if graph == "local" else 0.0, seed=0)


def make_workload(args, dev):
    """-> dict(name, ei_np (int64 [2,E] = the model's edge_index), mask_np, x (device tensor))"""
    from bridged_gnn_amd import synth, utils
    gen = torch.Generator(device=dev).manual_seed(0)
    if args.config == "c4":
        ei_np, mask_np = c4_graph(args.nodes, args.edges, args.graph)
        return {"name": f"C4 synthetic bridged graph ({args.graph})", "ei_np": ei_np, "mask_np": mask_np,
                "x": torch.randn(mask_np.shape[0], args.feat, device=dev, generator=gen)}
    if args.config == "c3":
        x, ei, y, m = synth.twitter_standin(seed=0)
        und = utils.to_undirected(torch.from_numpy(ei).to(dev), x.shape[0])       # run.sh:7 --to_undirected
        return {"name": "C3 Twitter_Graph stand-in (581 S + 20230 T, kNN + 0.9M random edges, undirected)",
                "ei_np": und.cpu().numpy(), "mask_np": m, "x": torch.from_numpy(x).to(dev)}
    if args.config == "c2":
        x, ei, y, m = synth.sync_rd_intra(n=10000, feat=args.feat, homophily=0.7, deg=10, k_cross=20, seed=0)
        und = utils.to_undirected(torch.from_numpy(ei).to(dev), x.shape[0])
        return {"name": "C2 Sync-RD_intra 10k nodes / 70% homophily (undirected)", "ei_np": und.cpu().numpy(), "mask_np": m,
                "x": torch.from_numpy(x).to(dev)}
    raise ValueError(args.config)


def build_model(args, dev):
    from bridged_gnn_amd.ktgnn import KTGNN_no_complement
    torch.manual_seed(0)
    model = KTGNN_no_complement(args.feat, args.classes, 2, args.hidden, root_weight=False, use_bn=True,
                                dim_share=args.feat, need_complement=False)
    g = torch.Generator().manual_seed(7)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    return model.to(dev).eval()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ------------------------------------------------------------------------------------------------ CPU legs (oracle)
def parity_report(ref, gpu_out, rows):
    """the timed forward's outputs against the oracle's: rtol 1e-5 + atol 1e-6 max|ref| (tests/conftest.py:assert_close)"""
    parity = {"rows_checked": int(rows), "checker": "oracle/oracle_c.c full forward on the same inputs",
              "bar": "rtol 1e-5 + atol 1e-6 * max|ref| per output"}
    worst, ok = 0.0, True
    for name, r, g in zip(("base", "target", "target_hat"), ref, gpu_out):
        g = g.cpu().numpy()
        err = np.abs(g.astype(np.float64) - r)
        tol = 1e-5 * np.abs(r) + 1e-6 * np.abs(r).max()
        parity[f"max_abs_err_{name}"] = float(err.max())
        worst = max(worst, float((err / tol).max()))
        ok = ok and bool((err <= tol).all())
    parity["max_err_over_tolerance"] = worst          # <= 1 : every element inside the bar
    parity["within_1e-5"] = ok
    return parity


def cpu_baseline(args, model, wl, gpu_out):
    """The oracle's C port (oracle/oracle_c.c, OpenMP) of the SAME eval forward on the SAME inputs, timed on the host
    cores.  Its first (warm-up) run doubles as the checker of the timed GPU forward's outputs (`parity`): every row of
    the three log-prob outputs.  When one forward takes the host longer than ~12 s (few cores) the timing falls back to
    a quarter-scale graph of the same generator and the full-size check is skipped."""
    from oracle import oracle_c as OC
    from oracle import oracle_np as O
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    ei, mask = wl["ei_np"], wl["mask_np"]
    quarter = None
    if args.config == "c4":                                  # probe the host's speed on the quarter-scale graph first
        n4 = max(args.nodes // 4, 1000)
        ei4, mask4 = c4_graph(n4, args.edges // 4, args.graph)
        x4 = np.random.default_rng(0).standard_normal((n4, args.feat)).astype(np.float32)
        rp4, col4, _ = O.dst_csr(ei4, mask4)
        OC.ktgnn_forward_eval(x4, rp4, col4, mask4, sd)
        t0 = time.perf_counter()
        OC.ktgnn_forward_eval(x4, rp4, col4, mask4, sd)
        quarter = (time.perf_counter() - t0, int(rp4[-1]), n4, (x4, rp4, col4, mask4))
    full = quarter is None or quarter[0] * 4 < 12.0
    parity = None
    if full:
        x = wl["x"].cpu().numpy()
        rowptr, col, _ = O.dst_csr(ei, mask)
        E = int(rowptr[-1])
        ref = OC.ktgnn_forward_eval(x, rowptr, col, mask, sd)          # warm-up run = the checker
        parity = parity_report(ref, gpu_out, mask.shape[0])
        run = lambda: OC.ktgnn_forward_eval(x, rowptr, col, mask, sd)
        sample = f"the same graph and inputs at full size (N={mask.shape[0]}, E'={E})"
    else:
        x4, rp4, col4, mask4 = quarter[3]
        E = quarter[1]
        run = lambda: OC.ktgnn_forward_eval(x4, rp4, col4, mask4, sd)
        sample = f"same generator at N={quarter[2]} nodes / E'={E} edges (1/4 scale; the host is too slow for full size)"
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    base = {"value": 4 * E / t, "unit": "edges/s", "cores": OC.num_threads(), "cpu": cpu_model(), "kind": "port",
            "what": "fused C/OpenMP port of the eval forward (oracle/oracle_c.c)",
            "sample": f"{sample}, 1 eval forward, median of 3: {t:.3f} s"}
    return base, parity


def cpu_baseline_torch(args, model):
    """SURVEY 8(d)'s CPU baseline: the op sequence the reference's PyG path executes (index_select / elementwise /
    index_add_ scatter, dense Linear), restated in plain CPU torch (oracle/oracle_torch.py, forward pinned to the
    reference's golden vectors), median of 5 runs (3 when a run takes the host > 6 s) on a 1/16-scale graph of the same
    generator.  Threads: at most 32 -- torch's CPU index_add_ gets SLOWER beyond that (33 s per forward on 256 threads)."""
    from oracle import oracle_torch as OT
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    n = max(args.nodes // 16, 1000)
    ei, mask = c4_graph(n, args.edges // 16, args.graph)
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((n, args.feat)).astype(np.float32))
    mo = torch.from_numpy(mask)
    e1, e2 = OT.graph_partition(torch.from_numpy(ei), mo)
    E = int(e1.shape[1] + e2.shape[1])
    ref = type(model)(args.feat, args.classes, 2, args.hidden, root_weight=False, use_bn=True, dim_share=args.feat)
    ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    ref.eval()
    conv = lambda c, xx: OT.adaptedconv(xx, mo, e1, e2, dict(c.named_parameters()))

    def fwd():
        with torch.no_grad():
            h = torch.relu(ref.bns[0](conv(ref.convs[0], x)))
            return (torch.log_softmax(conv(ref.clf_base, h), 1), torch.log_softmax(conv(ref.clf_target, h), 1),
                    torch.log_softmax(conv(ref.clf_target, ref.clf_transformer(h)), 1))
    t0 = time.perf_counter()
    fwd()
    runs = 5 if time.perf_counter() - t0 < 6.0 else 3
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fwd()
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    return {"value": 4 * E / t, "unit": "edges/s", "cores": torch.get_num_threads(), "cpu": cpu_model(), "kind": "port",
            "what": "reference-shaped torch op sequence (index_select / index_add_), oracle/oracle_torch.py",
            "sample": f"same generator at N={n} / E'={E} (1/16 scale), 1 eval forward, median of {runs}: {t:.3f} s"}


def knn_cpu_baselines(n, k=20, sample_queries=8192):
    """(i) the oracle's C port of the cosine kNN (OpenMP, exhaustive canonical fp64 scores + top-k) and (ii) the GEMM
    restatement SURVEY 8(d) asks for (torch sgemm + topk on the host), each on `sample_queries` of the query rows."""
    from bridged_gnn_amd import synth
    from oracle import oracle_c as OC
    nq = min(sample_queries, n)
    q = OC.l2_normalize_rows(synth.gaussian_embeddings(n, 128, seed=0)[:nq])
    c = OC.l2_normalize_rows(synth.gaussian_embeddings(n, 128, seed=1))
    OC.cosine_topk(q[:64], c, k)
    t0 = time.perf_counter()
    OC.cosine_topk(q, c, k)
    t = time.perf_counter() - t0
    port = {"value": float(nq) * n / t, "unit": "pairs/s", "cores": OC.num_threads(), "cpu": cpu_model(), "kind": "port",
            "what": "exhaustive canonical fp64 scores + top-k (oracle/oracle_c.c)",
            "sample": f"{nq} of the {n} query rows against all {n} candidates, d=128 k={k}: {t:.2f} s"}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    qt, ct = torch.from_numpy(q), torch.from_numpy(c).t().contiguous()
    (qt[:256] @ ct).topk(k, dim=1)
    t0 = time.perf_counter()
    for s in range(0, nq, 1024):
        torch.sigmoid(qt[s:s + 1024] @ ct).topk(k, dim=1, largest=True, sorted=False)      # main_bridged_graph.py:59-60
    t = time.perf_counter() - t0
    gemm = {"value": float(nq) * n / t, "unit": "pairs/s", "cores": torch.get_num_threads(), "cpu": cpu_model(), "kind": "port",
            "what": "GEMM restatement: per-node normalised q, torch sgemm + sigmoid + topk in batches of 1024 queries",
            "sample": f"{nq} of the {n} query rows against all {n} candidates: {t:.2f} s"}
    return port, gemm


def max_over_ranks(vals, dev):
    """MAX all-reduce of a few floats; on the host when the process group is gloo (the `--dist-backend gloo` rehearsal of
    the N>1 code path with several ranks on one GPU: RCCL refuses two ranks per device)."""
    on_host = torch.distributed.get_backend() == "gloo"
    t = torch.tensor(list(vals), dtype=torch.float64, device="cpu" if on_host else dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return [float(v) for v in t.tolist()]


# ------------------------------------------------------------------------------------------------ C5: kNN bridge
def knn_bench(args, dev, rank=0, world=1):
    """C5: cosine kNN bridge through the library's sharded entry point (bridge.sharded_cosine_topk_edges): query rows
    are sharded over the ranks, each rank normalises ITS slice of the candidates and one all_gather makes them whole
    (SURVEY 8(e)); the job time is the max over ranks.  The top-k call itself is timed with HIP events on the launch stream."""
    from bridged_gnn_amd import bridge, synth
    n = args.knn_n
    lo, hi = rank * n // world, (rank + 1) * n // world
    q = torch.from_numpy(synth.gaussian_embeddings(n, 128, seed=0)[lo:hi]).to(dev)
    c_mine = torch.from_numpy(synth.gaussian_embeddings(n, 128, seed=1)[lo:hi]).to(dev)   # this rank's share of the candidates
    ts, ks = [], []
    for it in range(4):
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ei, idx, val, nfb = bridge.sharded_cosine_topk_edges(q, c_mine, 20, rank=rank, world=world, query_base=lo, events=(s, e))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            dt = max_over_ranks([dt], dev)[0]
        ts.append(dt)
        ks.append(s.elapsed_time(e))
    t = float(np.median(ts[1:]))
    tk = float(np.median(ks[1:])) * 1e-3
    pairs = float(n) * float(n)
    pairs_rank = float(hi - lo) * float(n)
    pieces = int(os.environ.get("BGNN_KNN_FAST_PRODUCTS", "1"))     # bf16 MFMA products per pair term in the fast pass: cand_hi . query_hi
    issued = pairs_rank * 256 * pieces / tk / 1e12
    return {"workload": f"C5 cosine kNN {n}x{n} d=128 k=20 (normalise + score + top-k + coalesce)"
                        + (f", query rows sharded x{world}, one all_gather of the candidates" if world > 1 else ""),
            "pairs_per_s": pairs / t, "ms": t * 1e3, "fallback_rows": int(nfb[0].item()), "precise_pass_rows": int(nfb[1].item()), "edges_this_rank": int(ei.shape[1]),
            "roofline": {"bound": "mfma", "kernel": "bgnn_cosine_topk_f32 (bf16 split, head + main fast pass on bf16 MFMA, fp64 refine, precise / exhaustive stages)",
                         "achieved": issued, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": issued / BF16_MFMA_PEAK_TFLOPS,
                         "traffic": None, "ms_per_launch": tk * 1e3, "bf16_products_per_pair_term": pieces,
                         "algorithmic_tflops": pairs_rank * 256 / tk / 1e12,
                         "note": "achieved = bf16 MFMA flops actually issued (2 d x `bf16_products_per_pair_term` per pair) / time of the "
                                 "whole top-k call; the fast pass is bound by the shortlist upkeep (vector / LDS issue), not by the "
                                 "matrix pipe: one product per term instead of three cut the matrix work 3x and the time 1.6x"}}


# ------------------------------------------------------------------------------------------------ HIP-graph phase
def graph_phase(args, runner, barrier, use_dist, dev, rank, world, out, units):
    """Capture ONE forward (every launch of it, and at N>1 its RCCL collectives) into a HIP graph, check the replayed
    outputs against an eager forward on every rank, time the same K steps as replays.  `out` (rank 0's result line,
    already complete from the eager measurement) is switched to the replay numbers only when every rank verified its
    outputs and the replay is faster.  The ranks agree through the rendezvous STORE (host side, no collective) whether
    everybody captured before anybody replays, so a failed capture is a clean fallback to the eager line; a replay that
    stalls (a graph-launched collective is the one thing that cannot be rehearsed on a one-GPU box) trips the watchdog,
    which prints the eager line and exits NON-ZERO."""
    import threading
    done = threading.Event()

    def watchdog():
        if not done.wait(args.graph_timeout):
            if rank == 0:
                out["graph_replay_ms_per_step"] = f"stalled (> {args.graph_timeout:.0f} s), eager result kept"
                print(json.dumps(out), flush=True)
            os._exit(3)          # a stalled replay / collective is a failed run: the eager line is printed, the exit code says so
    threading.Thread(target=watchdog, daemon=True).start()
    note, g, got, ref, gdt = None, None, None, None, None
    try:
        with torch.no_grad():
            ref = [t.clone() for t in runner()[:3]]
            barrier()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                got = runner()[:3]
    except Exception as exc:                              # the eager measurement stands
        note = f"capture failed: {type(exc).__name__}: {str(exc)[:120]}"
    if use_dist and world > 1:
        from torch.distributed.distributed_c10d import _get_default_store
        store = _get_default_store()
        store.set(f"bgnn_capture_{rank}", "0" if note else "1")
        flags = [store.get(f"bgnn_capture_{r}").decode() for r in range(world)]      # blocks until every rank has posted
        if note is None and "0" in flags:
            note = "capture failed on another rank, eager result kept"
    if note is None:
        try:
            with torch.no_grad():
                for _ in range(3):
                    g.replay()
                barrier()
                ok = all(torch.allclose(a, b, rtol=1e-4, atol=1e-5) for a, b in zip(got, ref))
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    g.replay()
                barrier()
                gdt = time.perf_counter() - t0
            gdt, bad = max_over_ranks([gdt, 0.0 if ok else 1.0], dev) if use_dist else (gdt, 0.0 if ok else 1.0)
            if bad:
                note = "replayed outputs differ from the eager outputs, eager result kept"
        except Exception as exc:
            note = f"replay failed: {type(exc).__name__}: {str(exc)[:120]}"
    done.set()
    if out is None:
        return
    if note is not None:
        out["graph_replay_ms_per_step"] = note
        return
    graph_ms = gdt / args.steps * 1e3
    out["graph_replay_ms_per_step"] = graph_ms
    if graph_ms < out["ms_per_step"]:
        out["ms_per_step"] = graph_ms
        out["value"] = units / (graph_ms * 1e-3)
        out["config"]["execution"] = ("HIP-graph replay: the whole forward (all launches" +
                                      (" and the RCCL collectives" if use_dist else "") +
                                      ") captured once, outputs checked against the eager forward, K replays timed")


class AggTimer:
    """Time of the aggregation launch of width `D` (the roofline's kernel), with HIP events on the launch stream (torch's
    current stream is the stream handed to the C ABI).  Events around the call inside the forward loop are only right while the
    GPU is the bottleneck: on a small graph the host enqueues slower than the GPU drains and the interval fills with host
    time.  So the call seen in the timed loop is kept (its argument tensors stay alive) and REPLAYED back to back afterwards
    between one pair of events: the queue is then full and the interval is launch time."""

    def __init__(self, D):
        from bridged_gnn_amd import ops
        self.D, self.on, self.calls = D, False, []
        self.orig = ops.adaptedconv_aggregate
        ops.adaptedconv_aggregate = self

    def __call__(self, *a, **k):
        D = a[6] if len(a) > 6 else k["D"]
        if D == self.D and self.on:                 # `on` during ONE timed step: every launch of that width in it
            kk = dict(k)
            if kk.get("colsum") is not None:
                kk["colsum"] = torch.zeros_like(kk["colsum"])  # the replays must not add into the forward's accumulator
            self.calls.append((a, kk))
        return self.orig(*a, **k)

    def take_ms(self, steps, reps=20):
        """ms per step of the kept launches (the partitioned path aggregates a conv in two: interior rows, then boundary rows)"""
        calls, self.calls = self.calls, []
        if not calls:
            return float("nan")
        with torch.no_grad():
            for _ in range(3):
                for a, k in calls:
                    self.orig(*a, **k)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                for a, k in calls:
                    self.orig(*a, **k)
            e.record()
            torch.cuda.synchronize()
        return s.elapsed_time(e) / reps


def time_forward(runner, steps, warmup, barrier, timer):
    """-> (seconds of EXACTLY `steps` forwards between two barriers -- the driver's contract --, outputs, stats).  `stats`:
    `first_forward_ms` (the process's first forward on this workload: allocations, packed weights, the memoised input domain sums)
    and per-step times of a SECOND loop of max(steps, 20) forwards, each between its own pair of events (median / min / max:
    SURVEY 8(d) asks for the median of >= 20 forwards; the contract's figure is the mean over the bracketed loop)."""
    with torch.no_grad():
        # one untimed forward whose aggregation launches are kept for AggTimer.take_ms.  Its argument tensors stay alive, so
        # the allocator hands the following forwards other blocks: that one-time allocation must happen in the warm-up, not
        # in the timed loop (it cost 1-16 ms there, i.e. up to 0.8 ms per step of a 20-step run)
        timer.on = True
        barrier()
        t0 = time.perf_counter()
        out = runner()
        torch.cuda.synchronize()
        first_ms = (time.perf_counter() - t0) * 1e3
        timer.on = False
        for _ in range(max(warmup, 1)):
            out = runner()
        barrier()
        # rehearsal of the timed loop's own pattern (`steps` forwards enqueued back to back, then one sync): the first such burst
        # of a process makes the HIP runtime block the host once for 40-90 ms (seen at the 2nd iteration, 3 runs of 4, whatever the
        # kernels are) -- a one-time cost of the runtime, not of the path; untimed like the warm-up
        for _ in range(steps):
            out = runner()
        barrier()
        host, mem = [], []
        t0 = time.perf_counter()
        for _ in range(steps):
            out = runner()
            host.append(time.perf_counter())
            if os.environ.get("BENCH_TRACE"):
                mem.append((torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20))
        barrier()
        dt = time.perf_counter() - t0
        if os.environ.get("BENCH_TRACE"):
            print("host ms per iteration of the bracketed loop:", [round((b - a) * 1e3, 2) for a, b in zip([t0] + host[:-1], host)],
                  "final sync", round((t0 + dt - host[-1]) * 1e3, 2), "MiB allocated/reserved", mem, "allocator", {k: v for k, v in torch.cuda.memory_stats().items() if k in ("num_device_alloc", "num_device_free", "num_alloc_retries")}, file=sys.stderr)
        n2 = max(steps, 20)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n2)]
        for a, b in evs:
            a.record(); runner(); b.record()
        torch.cuda.synchronize()
        per = sorted(a.elapsed_time(b) for a, b in evs)
    stats = {"first_forward_ms": first_ms, "median_ms_per_step": float(np.median(per)), "min_ms_per_step": per[0],
             "max_ms_per_step": per[-1], "median_over_steps": n2}
    return dt, out, stats


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (see module docstring)")
    if args.dist_backend == "gloo":                     # rehearsal: ranks share the GPUs there are
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if args.dist_backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.config == "c5":                              # the kNN bridge alone
        knn = knn_bench(args, dev, rank, world)
        if rank == 0:
            out = {"metric": "knn_bridge_pairs_per_sec", "value": knn["pairs_per_s"], "unit": "pairs/s", "n_gpus": world,
                   "steps": 3, "warmup": 1, "ms_per_step": knn["ms"], "higher_is_better": True, "scaling": "strong",
                   "vs_baseline": None, "dtype": "f32 (bf16-piece MFMA shortlist, fp64 canonical re-score)", "data": "synthetic",
                   "config": {"workload": knn["workload"], "parallelism": "single" if world == 1 else f"query rows x{world}"},
                   "roofline": knn["roofline"], "fallback_rows": knn["fallback_rows"]}
            if world == 1 and not args.no_cpu:
                out["cpu_baseline"], out["cpu_baseline_gemm"] = knn_cpu_baselines(args.knn_n)
            print(json.dumps(out), flush=True)
        if use_dist:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return

    from bridged_gnn_amd.data import Data
    wl = make_workload(args, dev)
    ei_np, mask_np = wl["ei_np"], wl["mask_np"]
    N = mask_np.shape[0]
    model = build_model(args, dev)
    timer = AggTimer(args.hidden)

    if not use_dist:
        data = Data(x=wl["x"], edge_index=torch.from_numpy(ei_np).to(dev), central_mask=torch.from_numpy(mask_np).to(dev))
        t0 = time.perf_counter()
        csr = model._prepare(data)
        torch.cuda.synchronize()
        csr_first_ms = (time.perf_counter() - t0) * 1e3          # first call of the process: allocations + code-object load
        from bridged_gnn_amd import ops as _ops
        t0 = time.perf_counter()
        _ops.build_dst_csr(data.edge_index, N)
        torch.cuda.synchronize()
        csr_ms = (time.perf_counter() - t0) * 1e3                # the build itself (one-time per graph, cached by the model)
        Eprime = csr.num_edges
        runner = lambda: model(data)
        par = "single"
    else:
        from bridged_gnn_amd.dist import PartitionedKTGNN
        t0 = time.perf_counter()
        pk = PartitionedKTGNN(model, ei_np, mask_np, rank, world, dev, always_communicate=args.force_dist,
                              cache_input_halo=not args.no_input_halo_cache)
        torch.cuda.synchronize()
        csr_ms = csr_first_ms = (time.perf_counter() - t0) * 1e3   # (partition plan + the rank's two CSRs, first call)
        Eprime = pk.global_num_edges
        x_local = wl["x"][pk.owned_global].contiguous()          # the same seed on every rank
        wl["x"] = None
        runner = lambda: pk.forward(x_local)
        par = (f"dst-node-partition x{world}; per forward 1 small all-reduce (domain sums of h and T(h)) + 1 all_to_all of the "
               f"classifier stage's 48-byte halo rows; " +
               ("halo rows and all-reduced domain sums of the static input features resident (fetched once per version of x)"
                if not args.no_input_halo_cache else "hidden conv: domain-sum all-reduce + 512-byte halo rows exchanged every forward"))

    dt, gpu_out, fstats = time_forward(runner, args.steps, args.warmup, barrier, timer)
    if use_dist:
        dt = max_over_ranks([dt], dev)[0]
    ms_step = dt / args.steps * 1e3
    # per STEP: the partitioned path aggregates a conv in two launches (interior rows, then boundary rows)
    agg_ms = timer.take_ms(args.steps)
    gpu_out = [t.clone() for t in gpu_out[:3]]
    checksums = [float(t.double().sum().item()) for t in gpu_out]        # of THIS rank's rows

    # ---- the conservative roofline figure: the same hidden aggregation on the uniform-random variant of the graph (no
    #      neighbour reuse for the L2 to exploit; SURVEY 8(d) "also run the adversarial uniform-random variant")
    uniform = None
    if args.config == "c4" and args.graph == "local" and not use_dist and not args.no_uniform:
        ei_u, mask_u = c4_graph(args.nodes, args.edges, "uniform")
        model_u = build_model(args, dev)
        data_u = Data(x=wl["x"], edge_index=torch.from_numpy(ei_u).to(dev), central_mask=torch.from_numpy(mask_u).to(dev))
        e_u = model_u._prepare(data_u).num_edges
        k_u = max(args.steps // 2, 5)
        time_forward(lambda: model_u(data_u), k_u, 3, barrier, timer)[0]
        ms_u = timer.take_ms(k_u)
        b_u = agg_bytes(e_u, N, args.hidden)
        tr_u, src_u = pmc_traffic(args, world, "uniform")
        uniform = {"ms_per_launch": ms_u, "bytes_per_launch": b_u, "achieved": b_u / (ms_u * 1e-3) / 1e9,
                   "frac": b_u / (ms_u * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": tr_u, "traffic_source": src_u,
                   "fabric_frac": (tr_u / (ms_u * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr_u else None, "edges": e_u}
        del model_u, data_u

    # N=1 through the partitioned driver (no process group, world size 1): the path the N>1 runs take must cost what the plain
    # forward costs when there is nothing to exchange (VERDICT r2 #4d: within 3 %)
    part_n1 = None
    if not use_dist and args.config == "c4" and not args.no_partitioned_check:
        from bridged_gnn_amd.dist import PartitionedKTGNN
        t0 = time.perf_counter()
        pk1 = PartitionedKTGNN(model, ei_np, mask_np, 0, 1, dev)
        torch.cuda.synchronize()
        plan_s = time.perf_counter() - t0
        x1 = wl["x"][pk1.owned_global].contiguous()
        with torch.no_grad():
            for _ in range(5):
                o1 = pk1.forward(x1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                o1 = pk1.forward(x1)
            torch.cuda.synchronize()
            ms1 = (time.perf_counter() - t0) / args.steps * 1e3
            inv = torch.empty_like(pk1.owned_global); inv[pk1.owned_global] = torch.arange(N, device=dev)
            err = max(float((a[inv] - b).abs().max()) for a, b in zip(o1, gpu_out))
        part_n1 = {"ms_per_step": ms1, "vs_plain": ms1 / ms_step, "plan_build_s": plan_s, "max_abs_diff_vs_plain_outputs": err}
        del pk1, x1, o1
    train = None
    if args.train_steps < 0:
        args.train_steps = 5 if (args.config == "c4" and not use_dist) else 0
    if args.train_steps > 0 and not use_dist:
        # one optimisation step = train-mode forward (dropout, batch-stat BN) + HIP backward + Adam
        import torch.nn.functional as F
        gen = torch.Generator(device=dev).manual_seed(1)
        import copy
        tmodel = copy.deepcopy(model).train()      # the timed forward's weights stay as they are (parity / checksums refer to them)
        y = torch.randint(0, args.classes, (N,), device=dev, generator=gen)
        tm = torch.rand(N, device=dev, generator=gen) < 0.5
        cm = data.central_mask
        opt = torch.optim.Adam(tmodel.parameters(), lr=1e-3, weight_decay=5e-3, fused=True)   # same update, one launch
        tmt = tm & ~cm
        w_b, w_t = tm.float() / tm.sum(), tmt.float() / tmt.sum()
        yi = y[:, None]

        def nll(logp, w):
            # F.nll_loss(logp[mask], y[mask]) (main_graph_knowledge_transfer.py:44-54) without compacting the masked rows:
            # boolean indexing costs a nonzero() sync per term and torch's nll_loss reduces in a single block (0.24 ms each)
            return -(logp.gather(1, yi).squeeze(1) * w).sum()

        def step():
            opt.zero_grad()
            lb, lt, lth, _ = tmodel(data)
            loss = (2 * nll(lb, w_b) + nll(lt, w_t) + nll(lth, w_t)) / 4 \
                + F.kl_div(lth, lt, log_target=True, reduction="batchmean")
            loss.backward()
            opt.step()
            return loss
        step(); step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.train_steps):
            step()
        torch.cuda.synchronize()
        train = {"ms_per_train_step": (time.perf_counter() - t0) / args.train_steps * 1e3, "steps": args.train_steps,
                 "what": "SURVEY 8(f) rank 1, reported beside the headline (not part of `value`): train-mode forward (dropout 0.5, batch-stat "
                         "BatchNorm) + backward (HIP kernels: pull aggregation backward, fused BN/ReLU/dropout, prep / bf16x3 Gram / "
                         "W-stationary transform backward) + fused Adam, reference loss (main_graph_knowledge_transfer.py:44-54)"}
        if args.config in ("c2", "c3"):
            # the reference's own graph sizes: a step is ~170 short launches and host-bound -> one HIP graph per step
            try:
                gmodel = copy.deepcopy(model).train()
                gopt = torch.optim.Adam(gmodel.parameters(), lr=1e-3, weight_decay=5e-3, capturable=True)

                def loss_fn(o):
                    lb, lt, lth, _ = o
                    return (2 * nll(lb, w_b) + nll(lt, w_t) + nll(lth, w_t)) / 4 + F.kl_div(lth, lt, log_target=True, reduction="batchmean")
                run = gmodel.graphed_train_step(data, loss_fn, gopt)
                run(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10 * args.train_steps):
                    run()
                torch.cuda.synchronize()
                train["graph_replay_ms_per_train_step"] = (time.perf_counter() - t0) / (10 * args.train_steps) * 1e3
                train["graph_replay_loss"] = float(run().detach())
                del gmodel, gopt, run
            except Exception as e:                                   # reported, not fatal: the eager number stands
                train["graph_replay_ms_per_train_step"] = f"capture failed: {type(e).__name__}: {str(e)[:160]}"
        del tmodel, opt
    knn = knn_bench(args, dev, rank, world) if (not args.no_knn and args.config == "c4") else None   # every rank takes part
    out = None
    if rank == 0:
        n_local = N if not use_dist else len(pk.owned_global)
        e_local = Eprime if not use_dist else pk.local_num_edges
        bytes_launch = agg_bytes(e_local, n_local, args.hidden)
        achieved = bytes_launch / (agg_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(args, world, args.graph if args.config == "c4" else "fixed")
        gather_bytes = e_local * 4 * args.hidden                       # the row gathers alone (what the L2s serve)
        kern = f"agg_wide_fast_kernel<LF={max(16, 1 << (max(args.hidden - 1, 1) // 4).bit_length())}> (hidden AdaptedConv aggregation, D={args.hidden}; agg_wide_kernel when the launch is outside the 32-bit-addressing envelope)"
        tnote = ("PMC FETCH_SIZE x2 + WRITE_SIZE per launch from the committed rocprofv3 passes named in `traffic_source` (separate "
                 "--pmc runs of this command, NOT collected by this run); the counters sit on the fabric side of the L2s, so "
                 "Infinity-Cache hits are included (MI355X_MICROARCH: HBM)")
        # this workload's own launch: SURVEY 8(d)'s literal figure is `algorithmic_frac` -- NOT a bound on a graph with neighbour
        # locality (most gathers are L2 hits, the fraction exceeds 1); `fabric_frac` and `l2_gather_frac` are
        this = {"ms_per_launch": agg_ms, "bytes_per_launch": bytes_launch, "algorithmic_GBps": achieved,
                "algorithmic_frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "fabric_frac": (traffic / (agg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "l2_gather_frac": gather_bytes / (agg_ms * 1e-3) / 1e9 / L2_GATHER_GBS, "l2_gather_peak_GBps": L2_GATHER_GBS,
                "gathered_bytes_per_launch": gather_bytes}
        if uniform is not None:
            # the top-level fraction is the one that IS a bound: the same kernel on the uniform-random variant of the graph
            # (no neighbour reuse, algorithmic bytes ~ fabric bytes), measured in this run
            roof = {"bound": "hbm", "kernel": kern, "achieved": uniform["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": uniform["frac"], "traffic": uniform["traffic"], "traffic_source": uniform["traffic_source"],
                    "traffic_note": tnote, "bytes_per_launch": uniform["bytes_per_launch"], "ms_per_launch": uniform["ms_per_launch"],
                    "graph": f"uniform-random variant of the workload (E'={uniform['edges']}): no neighbour reuse, so SURVEY 8(d)'s byte "
                             "model B_agg(D) = E'(4D+4) + N(8D+4) + 4 bounds the launch; frac = bytes_per_launch / ms_per_launch / peak",
                    "fabric_frac": uniform["fabric_frac"],
                    "timed_workload_launch": this}
        else:
            a_frac = achieved / HBM_PEAK_GBS
            roof = {"bound": "hbm", "kernel": kern, "achieved": achieved if a_frac <= 1.0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": a_frac if a_frac <= 1.0 else None, "traffic": traffic, "traffic_source": traffic_src, "traffic_note": tnote,
                    "bytes_per_launch": bytes_launch, "ms_per_launch": agg_ms,
                    "graph": "the timed workload itself (no uniform-random variant was run); frac is null when the algorithmic "
                             "figure exceeds the peak (gathers served by the L2s): see timed_workload_launch",
                    "timed_workload_launch": this}
        out = {
            "metric": "aggregated_edges_per_sec_ktgnn_fwd", "value": 4 * Eprime / (ms_step * 1e-3), "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{wl['name']} N={N} E'={Eprime}, 2-layer KT-GNN eval fwd F={args.feat} hidden={args.hidden} C={args.classes}",
                       "parallelism": par, "csr_build_ms": csr_ms, "csr_build_first_call_ms": csr_first_ms,
                       "cached": ["by-destination CSR of the graph (built once, like the reference's cached graph_partition)",
                                  "per-domain column sums of the static input features x (memoised per tensor version; the "
                                  "HIP-graph replay recomputes them every forward)"] +
                                 (["input halo rows + all-reduced input domain sums"] if use_dist and not args.no_input_halo_cache else [])},
            "first_forward_ms": fstats["first_forward_ms"], "median_ms_per_step": fstats["median_ms_per_step"],
            "min_ms_per_step": fstats["min_ms_per_step"], "max_ms_per_step": fstats["max_ms_per_step"],
            "median_over_steps": fstats["median_over_steps"],
            "hidden_conv_edges_per_sec": e_local * world / (agg_ms * 1e-3),
            "roofline": roof,
            "output_checksums": {"what": "fp64 sums of the three log-prob outputs of the timed forward" +
                                         (" (this rank's rows)" if use_dist else ""),
                                 "base": checksums[0], "target": checksums[1], "target_hat": checksums[2]},
        }
        if part_n1 is not None:
            out["partitioned_path_n1"] = part_n1
        if use_dist:
            out["config"]["plan_build_s"] = csr_ms / 1e3
            out["config"]["world_size_seen"] = torch.distributed.get_world_size()
            out["config"]["dist_backend"] = torch.distributed.get_backend()
            out["config"]["halo_exchange"] = pk.halo.mode
        if knn is not None:
            out["knn"] = knn
        if train is not None:
            out["train"] = train
        if world == 1 and not use_dist and not args.no_cpu:
            out["cpu_baseline"], parity = cpu_baseline(args, model, wl, gpu_out)
            if parity is not None:
                out["parity"] = parity
            if args.config == "c4":
                out["cpu_baseline_torch"] = cpu_baseline_torch(args, model)
            if knn is not None:
                out["knn"]["cpu_baseline"], out["knn"]["cpu_baseline_gemm"] = knn_cpu_baselines(args.knn_n)
        out["eager_ms_per_step"] = ms_step
        out["config"]["execution"] = "eager launches"
    if not args.no_graph_replay:
        graph_phase(args, runner, barrier, use_dist, dev, rank, world, out, 4 * Eprime)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
