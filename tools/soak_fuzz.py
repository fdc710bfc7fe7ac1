"""Soak of the seeded parity fuzzers (aggregation, transform, cosine kNN vs the C oracle) over seeds the test suite does
not use: `python tools/soak_fuzz.py [first_seed [count]]` on a GPU box.  Round 1: seeds 100..259, 0 failures; round 2 (hub-row segments, lane-per-head kernels, kNN cascade): seeds 300..499, 0 failures; with the quad-cooperative exact re-score in the refine stage: seeds 500..699, 0 failures; after the scratch-clearing change (own zero-fill kernel): seeds 800..999, 0 failures; round 3 (stream / classifier-stage transform kernels, fast aggregation where the graph has no hub rows): seeds 1000..1199, 0 failures."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.chdir(os.path.join(ROOT, "tests"))
import pytest, importlib
import test_gpu_ktgnn as T, test_gpu_knn as K
bad = []
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 160
for seed in range(first, first + count):
    for name, fn in (("agg", T.test_aggregate_fuzz_vs_c_oracle), ("tr", T.test_transform_fuzz_vs_c_oracle), ("knn", K.test_cosine_topk_fuzz_bit_exact)):
        try:
            fn(seed)
        except pytest.skip.Exception:
            pass
        except BaseException as e:
            bad.append((name, seed, str(e)[:200])); print("FAIL", name, seed, str(e)[:300], flush=True)
print("soak done, failures:", len(bad))
