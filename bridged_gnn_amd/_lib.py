"""ctypes binding of `csrc/libbgnn_hip.so` (C ABI declared in include/bgnn.h).

There is NO fallback: if the library is absent or an operand is not a CUDA(HIP) tensor the call
raises.  PyTorch only supplies device memory and the current stream."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "csrc", "libbgnn_hip.so")
_lib = None

_P, _I64, _I32, _F32, _SZ, _INT = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_size_t, C.c_int

# name -> (restype, argtypes); mirrors include/bgnn.h one to one
SIGNATURES = {
    "bgnn_version": (_INT, []),
    "bgnn_source_hash": (C.c_char_p, []),
    "bgnn_error_string": (C.c_char_p, [_INT]),
    "bgnn_csr_workspace_bytes": (_SZ, [_I64, _I64]),
    "bgnn_build_dst_csr": (_INT, [_P, _I64, _I64, _INT, _P, _P, _P, _P, _P, _SZ, _P]),
    "bgnn_domain_sums_f64": (_INT, [_P, _I64, _I32, _I64, _P, _P, _P]),
    "bgnn_domain_sums_workspace_bytes": (_SZ, [_I32]),
    "bgnn_domain_sums_ws_f64": (_INT, [_P, _I64, _I32, _I64, _P, _P, _P, _SZ, _P]),
    "bgnn_domain_delta_f32": (_INT, [_P, _I32, _P, _P]),
    "bgnn_adaptedconv_transform_f32": (_INT, [_P, _I64, _I32, _I64, _P, _P, _I32, _I32, _P, _P, _P, _P,
                                               _P, _P, _P, _P, _I64, _I64, _P, _P]),
    "bgnn_adaptedconv_transform_sums_f32": (_INT, [_P, _I64, _I32, _I64, _P, _P, _I32, _I32, _P, _P, _P, _P,
                                                    _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P]),
    "bgnn_adaptedconv_transform_need_f32": (_INT, [_P, _I64, _I32, _I64, _P, _P, _I32, _I32, _P, _P, _P, _P,
                                                    _P, _P, _P, _P, _I64, _I64, _P, _P, _P]),
    "bgnn_classifier_stage_f32": (_INT, [_P, _I64, _I32, _I64, _P, _P, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64,
                                          _P, _P, _I32, _INT, _P, _P, _P, _P, _P, _P]),
    "bgnn_linear_narrow_transform_f32": (_INT, [_P, _I64, _I32, _I64, _P, _P, _I32, _INT, _P, _P, _P, _P, _P, _P]),
    "bgnn_narrow_transform_finish_f32": (_INT, [_P, _I64, _P, _P, _I32, _P, _P, _P, _P, _P, _P, _I64, _P, _P]),
    "bgnn_gram_workspace_bytes": (C.c_size_t, [_I32, _I32]),
    "bgnn_gram_f32": (_INT, [_P, _I64, _I32, _P, _I64, _I32, _I64, _P, _P, C.c_size_t, _P]),
    "bgnn_transform_bwd_prep_workspace_bytes": (C.c_size_t, [_I64, _I32]),
    "bgnn_transform_bwd_prep_f32": (_INT, [_P, _I64, _I64, _I32, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _I32, _I64, _P, _I64, _P, _P, C.c_size_t, _P]),
    "bgnn_bn_acc_doubles": (_I64, [_I32]),
    "bgnn_bn_relu_dropout_f32": (_INT, [_P, _I64, _I32, _I64, _P, _P, _F32, _INT, _F32, C.c_uint64, _P, _F32, _P, _P, _P, _I64, _P, _P]),
    "bgnn_bn_relu_dropout_bwd_f32": (_INT, [_P, _P, _I64, _I32, _I64, _I64, _P, _P, _P, _F32, _INT, _F32, C.c_uint64, _P, _P, _I64, _P, _P]),
    "bgnn_transform_bwd_consts_f32": (_INT, [_P, _P, _P, _P, _P, _I32, _I32, _P, _P, _P, _P]),
    "bgnn_transform_bwd_finish_f32": (_INT, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "bgnn_rowdot_f32": (_INT, [_P, _I64, _I64, _I32, _P, _I64, _I32, _P, _P]),
    "bgnn_linear_f32": (_INT, [_P, _I64, _I32, _I64, _P, _P, _I32, _INT, _P, _P, _P, _I64, _P]),
    "bgnn_adaptedconv_aggregate_f32": (_INT, [_P, _P, _I64, _P, _P, _P, _P, _P, _I64, _I64, _I32, _F32,
                                               _P, _I64, _P, _P, _P, _INT, _P, _INT, _I64, _I32, _P, _P, _P]),
    "bgnn_adaptedconv_aggregate_bounded_f32": (_INT, [_P, _P, _I64, _P, _P, _P, _P, _P, _I64, _I64, _I32, _F32,
                                                       _P, _I64, _P, _P, _P, _INT, _P, _INT, _I64, _I32, _P, _P, _I64, _I32, _P]),
    "bgnn_aggregate_hub_workspace_bytes": (C.c_size_t, [_I64, _I32, _I64]),
    "bgnn_adaptedconv_aggregate_hub_f32": (_INT, [_P, _P, _I64, _P, _P, _P, _P, _P, _I64, _I32, _F32, _P, _I64, _P, _P, _INT, _P, _I32, _P,
                                                   _P, _I32, _P, _I64, _P, _P, _P, _I64, _P, _P, C.c_size_t, _P]),
    "bgnn_adaptedconv_aggregate_bwd_f32": (_INT, [_P, _P, _I64, _P, _P, _P, _P, _P, _I64, _I64, _I32, _F32,
                                                   _P, _I64, _P, _P, _I64, _P, _P, _P, _P, _P]),
    "bgnn_aggregate_bwd_pull_workspace_bytes": (C.c_size_t, [_I64, _I64, _I64]),
    "bgnn_adaptedconv_aggregate_bwd_pull_f32": (_INT, [_P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I32, _F32,
                                                        _P, _I64, _P, _P, _I64, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "bgnn_aggregate_bwd_pull_hub_workspace_bytes": (C.c_size_t, [_I64, _I64, _I64, _I64, _I64]),
    "bgnn_adaptedconv_aggregate_bwd_pull_hub_f32": (_INT, [_P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I32, _F32,
                                                            _P, _I64, _P, _P, _I64, _P, _P, _P, _P, _I32,
                                                            _P, _I64, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, _I64, _P, C.c_size_t, _P]),
    "bgnn_aggregate_heads_bwd_workspace_bytes": (C.c_size_t, [_I64, _I64, _I32]),
    "bgnn_adaptedconv_aggregate_heads_bwd_f32": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _F32,
                                                         _P, _P, _P, _INT, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "bgnn_aggregate_heads_bwd_hub_workspace_bytes": (C.c_size_t, [_I64, _I64, _I32, _I64, _I64]),
    "bgnn_adaptedconv_aggregate_heads_bwd_hub_f32": (_INT, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _F32,
                                                             _P, _P, _P, _INT, _P, _P, _P, _P, _I32,
                                                             _P, _I64, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, _I64, _P, C.c_size_t, _P]),
    "bgnn_l2_normalize_rows_f32": (_INT, [_P, _I64, _I32, _F32, _P, _P]),
    "bgnn_topk_workspace_bytes": (_SZ, [_I64, _I64, _I32]),
    "bgnn_cosine_topk_f32": (_INT, [_P, _P, _I64, _I64, _I32, _I32, _INT, _P, _P, _P, _P, _SZ, _P]),
    "bgnn_mlp_pair_topk_f32": (_INT, [_P, _P, _P, _P, _P, _F32, _I64, _I64, _I32, _I32, _INT, _P, _P, _P,
                                       _P, _SZ, _P]),
    "bgnn_topk_edges_i64": (_INT, [_P, _I64, _I32, _I64, _I64, _P, _P]),
    "bgnn_topk_edges_coalesced_workspace_bytes": (_SZ, [_I64, _I32]),
    "bgnn_topk_edges_coalesced_i64": (_INT, [_P, _I64, _I32, _I64, _I64, _I64, _P, _P, _SZ, _P]),
    "bgnn_gather_rows_f32": (_INT, [_P, _I64, _I64, _P, _I64, _I32, _P, _I64, _P]),
    "bgnn_coalesce_workspace_bytes": (_SZ, [_I64]),
    "bgnn_coalesce_i64": (_INT, [_P, _I64, _I64, _P, _P, _SZ, _P]),
}


def source_hash():
    """sha256 (first 16 hex digits) over the library's sources in the Makefile's order -- the same digest the
    Makefile bakes into the library as `bgnn_source_hash()`; a mismatch means the .so is older than the sources."""
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    h = hashlib.sha256()
    for name in _HASHED_SOURCES:
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


# keep in step with HASHED in csrc/Makefile
_HASHED_SOURCES = ("bgnn_api.hip", "bgnn_csr.hip", "bgnn_transform.hip", "bgnn_transform_stream.hip", "bgnn_transform_cls.hip", "bgnn_aggregate.hip", "bgnn_aggregate_bwd.hip",
                   "bgnn_aggregate_bwd_fast.hip", "bgnn_knn.hip", "bgnn_gram.hip", "bgnn_norm.hip", "bgnn_common.h", "bgnn_transform_params.h", "bgnn_aggregate_bwd_params.h", os.path.join("..", "..", "include", "bgnn.h"))


def _sidecar_hash():
    try:
        with open(os.path.join(_HERE, "csrc", "libbgnn_hip.srchash")) as f:
            return f.read().strip()
    except OSError:
        return None


# the ABI revision this binding was written for (include/bgnn.h: BGNN_VERSION); checked against the loaded library
ABI_VERSION = 113


def _make():
    """Rebuild the library under an exclusive file lock: in a multi-rank launch after a source edit every rank finds the
    stale sidecar at once, and unserialised `make` runs in one directory corrupt the objects / hand a rank a half-written
    .so.  The first rank in builds (the Makefile links to a temporary name and renames it into place, sidecar last); the
    others wait on the lock, re-check the sidecar and find nothing left to do.  A failing build raises with make's output."""
    import fcntl
    import subprocess
    csrc = os.path.join(_HERE, "csrc")
    with open(os.path.join(csrc, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if os.path.exists(SO_PATH) and _sidecar_hash() == source_hash():
                return                                     # another process built it while this one waited
            r = subprocess.run(["make", "-C", csrc, "-j4", "-s"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"building {SO_PATH} failed (make exit code {r.returncode}):\n{r.stdout[-4000:]}")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _load():
    l = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype, fn.argtypes = res, args
    return l


def lib():
    """Load libbgnn_hip.so; raises (loudly) when it has not been built or was built from other sources than the
    ones next to it (a stale .so would otherwise load silently)."""
    global _lib
    if _lib is None:
        have_hipcc = os.path.exists("/opt/rocm/bin/hipcc")
        if have_hipcc and (not os.path.exists(SO_PATH) or _sidecar_hash() != source_hash()):
            # building the HIP library is not a fallback: same kernels, compiled where they were missing or stale
            # (the sidecar file lets the staleness be seen BEFORE the library is mapped into the process)
            _make()
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C bridged_gnn_amd/csrc`).  bridged_gnn_amd has no CPU fallback.")
        want = source_hash()
        l = _load()
        got = l.bgnn_source_hash().decode()
        if got != want:
            raise RuntimeError(
                f"{SO_PATH} was built from other sources (library {got}, tree {want}): rebuild it with "
                "`make -C bridged_gnn_amd/csrc` (a loaded library cannot be replaced inside this process)")
        if l.bgnn_version() != ABI_VERSION:
            raise RuntimeError(f"{SO_PATH} exports ABI revision {l.bgnn_version()}, this binding was written for {ABI_VERSION}")
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().bgnn_error_string(int(rc))
        raise RuntimeError(f"{what} failed: {msg.decode() if msg else rc} (code {rc})")


def _check_device(t):
    if not t.is_cuda:
        raise RuntimeError("bridged_gnn_amd ops need CUDA(HIP) tensors; there is no CPU path "
                           f"(got a {t.device} tensor)")
    if t.device.index != torch._C._cuda_getDevice():
        # launches go to the CURRENT device's stream: a tensor of another GPU would hand the kernel foreign pointers
        raise RuntimeError(f"tensor lives on {t.device} but the current device is cuda:{torch._C._cuda_getDevice()}; "
                           "wrap the call in `with torch.cuda.device(t.device):` (one process drives one GPU here)")


def ptr(t):
    """Device pointer of a tensor (None -> NULL); refuses host tensors and tensors of a device that is not current."""
    if t is None:
        return None
    _check_device(t)
    if not t.is_contiguous():
        raise RuntimeError("bridged_gnn_amd ops need contiguous tensors")
    return C.c_void_p(t.data_ptr())


def ptr_rows(t):
    """Device pointer of a 2-D row-strided view (unit column stride, e.g. a column slice of an interleaved table)."""
    if t is None:
        return None
    _check_device(t)
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError("expected a 2-D tensor with unit column stride")
    return C.c_void_p(t.data_ptr())


def raw_stream():
    """hipStream_t of torch's current stream as an int.  `torch.cuda.current_stream()` resolves the device through
    `torch.cuda.is_available()` (an os.getenv per call, ~25 us): too slow for a forward made of ~25 launches."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def stream():
    return C.c_void_p(raw_stream())
