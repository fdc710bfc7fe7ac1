#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_r02.sh into the tracked evidence under profiles/<round>/:
  python tools/pmc_summary.py gpurun_out/<tag> profiles/r02
* kernel_stats_<name>.csv           the --stats summary tables (copied)
* counters_<pass>.csv               raw per-dispatch counter rows of the kernels of interest (copied, filtered by kernel name)
* pmc_summary.json                  per pass and kernel: dispatches, mean of every counter
* pmc_traffic_c4_{local,uniform}.json   HBM bytes per launch of the hidden aggregation (bench.py reads these)
Corrections (MI355X_MICROARCH.md, HBM / rocprofv3): FETCH_SIZE / WRITE_SIZE are reported in KB (1024 B); FETCH_SIZE counts
exactly half of the bytes of wide (16 B per lane) reads on gfx950 -> x2; WRITE_SIZE is exact."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

KEEP = ("agg_wide", "cls_stage", "agg_heads", "agg_kernel", "agg_bwd", "gram_", "bn_", "colstats", "transform_bwd", "transform_wreg_kernel", "transform_stream", "transform_gemm_kernel", "transform_skinny_kernel",
        "domain_sums_kernel", "cosine_pass1", "knn_", "refine_kernel", "normalize_rows_kernel", "narrow_finish_kernel")


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    summary = {}
    for d in sorted(glob.glob(os.path.join(src, "*"))):
        if not os.path.isdir(d):
            continue
        tag = os.path.basename(d)
        for f in glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(dst, f"kernel_stats_{tag}.csv"))
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            rows = list(csv.DictReader(open(f)))
            keep = [r for r in rows if any(k in r["Kernel_Name"] for k in KEEP)]
            if keep:
                with open(os.path.join(dst, f"counters_{tag}.csv"), "w", newline="") as out:
                    w = csv.DictWriter(out, fieldnames=list(keep[0].keys()))
                    w.writeheader()
                    w.writerows(keep)
            acc = defaultdict(lambda: defaultdict(list))
            for r in keep:
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            summary[tag] = {k: {c: {"mean": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()} for k, cs in acc.items()}
    json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
    for g in ("local", "uniform"):
        try:
            fk = next(k for k in summary[f"fetch_{g}"] if k.startswith("agg_wide"))
            fetch = summary[f"fetch_{g}"][fk]["FETCH_SIZE"]["mean"]
            write = summary[f"write_{g}"][fk]["WRITE_SIZE"]["mean"]
        except (KeyError, StopIteration):
            continue
        rec = {"kernel": f"{fk} hidden AdaptedConv aggregation, C4 {g} graph", "fetch_size_kb": fetch, "write_size_kb": write,
               "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
               "correction": "FETCH_SIZE x2 (gfx950 wide-read undercount, MI355X_MICROARCH.md HBM section), WRITE_SIZE exact; KB = 1024 B",
               "source": f"counters_fetch_{g}.csv + counters_write_{g}.csv (rocprofv3 --pmc, tools/profile_r03.sh)",
               "workload": {"nodes": 1000000, "edges": 20000000, "hidden": 128, "graph": g}}
        tcc = summary.get(f"tcc_{g}", {}).get(fk)
        if tcc and "TCC_HIT_sum" in tcc:
            h, m = tcc["TCC_HIT_sum"]["mean"], tcc["TCC_MISS_sum"]["mean"]
            rec["tcc_hit_rate"] = h / (h + m)
        json.dump(rec, open(os.path.join(dst, f"pmc_traffic_c4_{g}.json"), "w"), indent=1)
    print(json.dumps({k: list(v.keys()) for k, v in summary.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
