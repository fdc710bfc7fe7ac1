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
    "bgnn_linear_narrow_transform_f32": (_INT, [_P, _I64, _I32, _I64, _P, _P, _I32, _INT, _P, _P, _P, _P, _P, _P]),
    "bgnn_narrow_transform_finish_f32": (_INT, [_P, _I64, _P, _P, _I32, _P, _P, _P, _P, _P, _P, _I64, _P, _P]),
    "bgnn_gram_workspace_bytes": (C.c_size_t, [_I32, _I32]),
    "bgnn_gram_f32": (_INT, [_P, _I64, _I32, _P, _I64, _I32, _I64, _P, _P, C.c_size_t, _P]),
    "bgnn_transform_bwd_prep_f32": (_INT, [_P, _I64, _I64, _I32, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _I32, _P, _P]),
    "bgnn_rowdot_f32": (_INT, [_P, _I64, _I64, _I32, _P, _I64, _I32, _P, _P]),
    "bgnn_linear_f32": (_INT, [_P, _I64, _I32, _I64, _P, _P, _I32, _INT, _P, _P, _P, _I64, _P]),
    "bgnn_adaptedconv_aggregate_f32": (_INT, [_P, _P, _I64, _P, _P, _P, _P, _P, _I64, _I64, _I32, _F32,
                                               _P, _I64, _P, _P, _P, _INT, _P, _INT, _I64, _I32, _P, _P, _P]),
    "bgnn_adaptedconv_aggregate_bwd_f32": (_INT, [_P, _P, _I64, _P, _P, _P, _P, _P, _I64, _I64, _I32, _F32,
                                                   _P, _I64, _P, _P, _I64, _P, _P, _P, _P, _P]),
    "bgnn_aggregate_bwd_pull_workspace_bytes": (C.c_size_t, [_I64, _I64, _I64]),
    "bgnn_adaptedconv_aggregate_bwd_pull_f32": (_INT, [_P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I32, _F32,
                                                        _P, _I64, _P, _P, _I64, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "bgnn_l2_normalize_rows_f32": (_INT, [_P, _I64, _I32, _F32, _P, _P]),
    "bgnn_topk_workspace_bytes": (_SZ, [_I64, _I64, _I32]),
    "bgnn_cosine_topk_f32": (_INT, [_P, _P, _I64, _I64, _I32, _I32, _INT, _P, _P, _P, _P, _SZ, _P]),
    "bgnn_mlp_pair_topk_f32": (_INT, [_P, _P, _P, _P, _P, _F32, _I64, _I64, _I32, _I32, _INT, _P, _P, _P,
                                       _P, _SZ, _P]),
    "bgnn_topk_edges_i64": (_INT, [_P, _I64, _I32, _I64, _I64, _P, _P]),
    "bgnn_gather_rows_f32": (_INT, [_P, _I64, _I64, _P, _I64, _I32, _P, _I64, _P]),
    "bgnn_coalesce_workspace_bytes": (_SZ, [_I64]),
    "bgnn_coalesce_i64": (_INT, [_P, _I64, _I64, _P, _P, _SZ, _P]),
}


def lib():
    """Load libbgnn_hip.so; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH) and os.path.exists("/opt/rocm/bin/hipcc"):
            # building the HIP library is not a fallback: same kernels, compiled where they were missing
            import subprocess
            subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4", "-s"], check=False)
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C bridged_gnn_amd/csrc`).  bridged_gnn_amd has no CPU fallback.")
        l = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().bgnn_error_string(int(rc))
        raise RuntimeError(f"{what} failed: {msg.decode() if msg else rc} (code {rc})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL); refuses host tensors."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("bridged_gnn_amd ops need CUDA(HIP) tensors; there is no CPU path "
                           f"(got a {t.device} tensor)")
    if not t.is_contiguous():
        raise RuntimeError("bridged_gnn_amd ops need contiguous tensors")
    return C.c_void_p(t.data_ptr())


def ptr_rows(t):
    """Device pointer of a 2-D row-strided view (unit column stride, e.g. a column slice of an interleaved table)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("bridged_gnn_amd ops need CUDA(HIP) tensors; there is no CPU path "
                           f"(got a {t.device} tensor)")
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError("expected a 2-D tensor with unit column stride")
    return C.c_void_p(t.data_ptr())


def raw_stream():
    """hipStream_t of torch's current stream as an int.  `torch.cuda.current_stream()` resolves the device through
    `torch.cuda.is_available()` (an os.getenv per call, ~25 us): too slow for a forward made of ~25 launches."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def stream():
    return C.c_void_p(raw_stream())
