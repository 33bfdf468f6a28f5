"""ctypes binding of libtrinerflet_hip.so (include/trinerflet_hip.h).

The product path has NO fallback: if the library is missing, or a call returns a HIP error,
this raises.  Nothing here imports oracle/.
"""
import ctypes as C
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# TNL_LIB_PATH: another build of the same library (kernel experiments: tools/build_variant.py); the product loads the
# in-tree build
LIB_PATH = os.environ.get("TNL_LIB_PATH") or os.path.join(_HERE, "libtrinerflet_hip.so")
HEADER = os.path.join(_HERE, "..", "include", "trinerflet_hip.h")

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: run `python -m trinerflet_amd.build` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the hot path.")
        _lib = C.CDLL(LIB_PATH)
        for name in ("tnl_march_rays_train_workspace", "tnl_march_rays_train_workspace_rec", "tnl_field_packed_bytes"):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = C.c_uint32
        for name in ("tnl_plane_grad_binned_workspace", "tnl_permute_index", "tnl_field_backward_workspace",
                     "tnl_field_feats_save_bytes", "tnl_abs_mean_workspace"):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = C.c_uint64
    return _lib


def declared_symbols():
    """Names of every TNL_API function declared in include/trinerflet_hip.h."""
    text = open(HEADER).read()
    return sorted(set(re.findall(r"TNL_API\s+[\w\s\*]+?\b(tnl_\w+)\s*\(", text)))


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def check(err, what):
    if err != 0:
        raise RuntimeError(f"{what} failed: hipError {err}")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("trinerflet_amd: tensor is not on a HIP device (no CPU fallback exists)")


u32 = C.c_uint32
f32 = C.c_float
i32 = C.c_int
u64 = C.c_uint64


def roi_array(roi):
    """10 host int32 {ox0,ox1,ox2, oy0,oy1,oy2, rw, rh, spp, s0} for the *_roi entry points (None -> NULL)."""
    if roi is None:
        return None
    vals = [int(v) for v in roi]
    assert len(vals) == 10
    return (C.c_int32 * 10)(*vals)
