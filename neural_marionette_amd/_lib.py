"""ctypes binding of libnm355.so (C ABI: include/nm355.h).

The HIP library is the product; there is no CPU or PyTorch fallback.  Importing this
module never needs a GPU (so the CPU test-suite can check the exported symbols), but
every compute entry point raises if the library is missing or a call fails.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NM355_LIB_PATH") or os.path.join(_HERE, "libnm355.so")      # (override: A/B builds of the same library, tools/ab_*.sh)

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


class NmConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("grid_size", C.c_int32), ("nkeypoints", C.c_int32),
        ("nlatent", C.c_int32), ("nhidden", C.c_int32), ("nneighbor", C.c_int32),
        ("gaussian_sigma", C.c_float), ("sep_sigma", C.c_float),
        ("vol_fit_chamfer", C.c_int32), ("use_graph_traj", C.c_int32),
    ]


class NmNamedTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


_P = C.c_void_p   # device pointers travel as plain integers
_I = C.c_int32
_F = C.c_float

# name -> (restype, argtypes); must list every symbol include/nm355.h declares
SIGNATURES = {
    "nm_abi_version": (C.c_int, []),
    "nm_last_error": (C.c_char_p, []),
    "nm_abi_selftest_throw": (C.c_int, [_I]),
    "nm_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(NmConfig)]),
    "nm_ctx_destroy": (C.c_int, [C.c_void_p]),
    "nm_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nm_ctx_check_nonfinite": (C.c_int, [C.c_void_p]),
    "nm_ctx_set_weights": (C.c_int, [C.c_void_p, C.POINTER(NmNamedTensor), _I]),
    "nm_workspace_bytes": (C.c_size_t, [C.c_void_p, _I, _I]),
    "nm_ctx_memory": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "nm_detector_forward": (C.c_int, [C.c_void_p, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "nm_detector_keypoints": (C.c_int, [C.c_void_p, _P, _I, _I, _I, _P, _P, _P, _P]),
    "nm_forward_fused": (C.c_int, [C.c_void_p, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nm_decode_from_keypoints": (C.c_int, [C.c_void_p, _P, _P, _P, _I, _I, _P]),
    "nm_get_affinity": (C.c_int, [C.c_void_p, _P]),
    "nm_ctx_set_affinity_ver": (C.c_int, [C.c_void_p, _I]),
    "nm_ctx_set_gaussian_cat": (C.c_int, [C.c_void_p, _I]),
    "nm_ctx_set_learnable_sigma": (C.c_int, [C.c_void_p, _I]),
    "nm_voxelize_clip": (C.c_int, [C.c_void_p, _P, _I, C.c_int64, C.c_double, _P, _P]),
    "nm_eval_voxel_chamfer": (C.c_int, [C.c_void_p, _P, _P, _I, _I, _I, _P]),
    "nm_eval_semantic": (C.c_int, [C.c_void_p, _P, _P, _I, _I, _I, _P, _P]),
    "nm_vrnn_set_tree": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p]),
    "nm_vrnn_offsets": (C.c_int, [C.c_void_p, _P, _I, _I, _P]),
    "nm_vrnn_encode": (C.c_int, [C.c_void_p, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "nm_vrnn_encode_train": (C.c_int, [C.c_void_p, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "nm_vrnn_encode_backward": (C.c_int, [C.c_void_p, _P, C.POINTER(NmNamedTensor), _I]),
    "nm_adam_step": (C.c_int, [C.c_void_p, _P, _P, _P, _P, C.c_int64, _I, _F, _F, _F, _F]),
    "nm_ctx_set_training": (C.c_int, [C.c_void_p, _I]),
    "nm_detector_forward_train": (C.c_int, [C.c_void_p, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "nm_detector_backward": (C.c_int, [C.c_void_p, _P, C.POINTER(NmNamedTensor), _I]),
    "nm_ctx_set_backward_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nm_adam_step_multi": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_int64), _I, _I, _F, _F, _F, _F]),
    "nm_adam_step_multi_ok": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_int64), _I, _I, _F, _F, _F, _F, _P]),
    "nm_vrnn_generate": (C.c_int, [C.c_void_p, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "nm_vrnn_rollout": (C.c_int, [C.c_void_p, _P, _P, _P, _I, _I, _P, _P]),
    "nm_vrnn_step": (C.c_int, [C.c_void_p, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "nm_rows_argmin_dist": (C.c_int, [C.c_void_p, _P, _P, _I, _I, _I, _P, _P]),
    "nm_vrnn_mlp": (C.c_int, [C.c_void_p, _I, _P, _I, _P]),
    "nm_vrnn_gru": (C.c_int, [C.c_void_p, _P, _P, _I, _P]),
    "nm_vrnn_fk": (C.c_int, [C.c_void_p, _P, _P, _I, _P, _P]),
    "nm_op_conv3d": (C.c_int, [C.c_void_p, _P, _I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _I, _I, _I, _I, _P,
                                _I, _P, _P, _P, _P, _I]),
    "nm_op_conv5_occ": (C.c_int, [C.c_void_p, _P, _I, _I, _P, _P, _I, _P, _I, _P, _P, _P, _P]),
    "nm_op_convT2": (C.c_int, [C.c_void_p, _P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _I, _P, _P, _P, _P]),
    "nm_op_apply2": (C.c_int, [C.c_void_p, _P, _P, _P, _F, _P, _P, _P, _F, _I, _I, _I, _P]),
    "nm_op_upsample2": (C.c_int, [C.c_void_p, _P, _I, _I, _I, _I, _I, _P]),
    "nm_op_pack_input": (C.c_int, [C.c_void_p, _P, _I, _I, _I, _I, _P]),
    "nm_op_cl_to_ncdhw": (C.c_int, [C.c_void_p, _P, _I, _I, _I, _P]),
    "nm_op_conv3d_backward": (C.c_int, [C.c_void_p, _P, _I, _I, _I, _I, _I, _P, _P, _F, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P]),
    "nm_op_conv5_occ_backward": (C.c_int, [C.c_void_p, _P, _I, _I, _I, _P, _P, _P, _I]),
    "nm_op_convT2_backward": (C.c_int, [C.c_void_p, _P, _I, _I, _I, _I, _I, _P, _P, _F, _P, _I, _I, _P, _P, _P, _P]),
    "nm_op_gn_backward": (C.c_int, [C.c_void_p, _P, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P, _P]),
    "nm_set_conv_mode": (C.c_int, [C.c_void_p, _I]),
    "nm_get_conv_mode": (C.c_int, [C.c_void_p]),
    "nm_op_set_storage16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "nm_prof_enable": (C.c_int, [C.c_void_p, _I]),
    "nm_prof_read": (C.c_int, [C.c_void_p, _I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "nm_prof_kernel_name": (C.c_char_p, [_I]),
    "nm_host_linspace": (C.c_int, [_I, c_float_p]),
}

_lib: Optional[C.CDLL] = None


NM_ERR_ARG, NM_ERR_HIP, NM_ERR_STATE, NM_ERR_UNSUPPORTED, NM_ERR_RANGE, NM_ERR_INTERNAL = -1, -2, -3, -4, -5, -6     # include/nm355.h


class NmError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen libnm355.so and bind every entry point; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NmError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C neural_marionette_amd/csrc).  There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise NmError(f"libnm355.so does not export {name}")
        fn.restype = res
        fn.argtypes = args
    if lib.nm_abi_version() != 1:
        raise NmError("libnm355.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().nm_last_error()
        raise NmError(f"nm355 {what} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous fp32/int32 CUDA(HIP) tensor (None -> NULL); bfloat16 for the op-level entry points under
    nm_op_set_storage16 (unit parity of the 16-bit storage kernels)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NmError("nm355 needs device tensors (HIP); got a CPU tensor — there is no CPU fallback")
    if not t.is_contiguous():
        raise NmError("nm355 needs contiguous tensors")
    if t.dtype not in (torch.float32, torch.int32, torch.float64, torch.bfloat16):
        raise NmError(f"nm355 needs fp32/int32 (fp64 for point clouds; bfloat16 for the 16-bit storage op tests) tensors, got {t.dtype}")
    return t.data_ptr()


class Context:
    """Owns one nm_ctx (one device, one stream)."""

    def __init__(self, cfg: NmConfig):
        self.lib = load()
        self.handle = C.c_void_p()
        self.cfg = cfg
        check(self.lib.nm_ctx_create(C.byref(self.handle), C.byref(cfg)), "ctx_create")
        self.device = torch.device("cuda", cfg.device)

    def bind_stream(self) -> None:
        s = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.nm_ctx_set_stream(self.handle, C.c_void_p(s)), "set_stream")

    def close(self) -> None:
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.nm_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
