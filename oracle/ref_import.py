"""ORACLE / TEST INFRASTRUCTURE -- build-container only.

Imports the reference's own Python (read-only at /root/reference) under the local
`oracle/shim` restatement of the absent third-party packages, so `gen_golden.py` can emit golden
vectors.  /root/reference does not exist on the GPU box; nothing in tests -m gpu, smoke() or
bench.py imports this module.  Procedure follows SURVEY.md section 8(c).
"""
import os
import sys
import types

REF_ROOT = "/root/reference"
REF_CODE = os.path.join(REF_ROOT, "Bridged-GNN")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shim")


def reference_available():
    return os.path.isdir(REF_CODE)


def import_reference():
    """Returns (KTGNN_module, models_module, main_bridged_graph_module)."""
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True  # the reference tree is read-only
    for p in (_SHIM, REF_CODE, os.path.join(REF_CODE, "models")):
        if p not in sys.path:
            sys.path.insert(0, p)
    if "datasets" not in sys.modules or not hasattr(sys.modules["datasets"], "prepare_datasets"):
        # datasets.py:134-139 loads absent raw data at import time -> stub the module
        stub = types.ModuleType("datasets")
        stub.prepare_datasets = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("datasets absent"))
        sys.modules["datasets"] = stub
    cwd = os.getcwd()
    os.chdir(REF_CODE)  # drivers do sys.path.append('./models')
    try:
        import KTGNN  # noqa
        import models  # noqa
        import main_bridged_graph  # noqa
    finally:
        os.chdir(cwd)
    return KTGNN, models, main_bridged_graph
