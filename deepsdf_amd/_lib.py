"""ctypes binding of libdsdf_hip.so (include/dsdf.h).  Fails loudly when the library is missing: there is
no CPU fallback anywhere in this package."""
import ctypes as C
import os

from .build import LIB

MAX_LAYERS = 16
ABI_VERSION = 15
MAX_BUCKETS = 8


class DsdfNet(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("latent_size", C.c_int32), ("geom_dim", C.c_int32),
                ("in_dim", C.c_int32 * MAX_LAYERS), ("out_dim", C.c_int32 * MAX_LAYERS),
                ("weight_norm_mask", C.c_uint32), ("dropout_mask", C.c_uint32), ("skip_mask", C.c_uint32),
                ("dropout_p", C.c_float), ("use_tanh", C.c_int32), ("fwd_bf16", C.c_int32),
                ("latent_dropout", C.c_int32), ("xyz_in_all", C.c_int32), ("ln_param_mask", C.c_uint32),
                ("gemm_split", C.c_int32)]


class DsdfParamLayout(C.Structure):
    _fields_ = [("total", C.c_int64), ("bias_off", C.c_int64 * MAX_LAYERS), ("g_off", C.c_int64 * MAX_LAYERS),
                ("v_off", C.c_int64 * MAX_LAYERS), ("ln_w_off", C.c_int64 * MAX_LAYERS), ("ln_b_off", C.c_int64 * MAX_LAYERS)]


class DsdfBatch(C.Structure):
    _fields_ = [("seg_scene", C.c_void_p), ("seg_offset", C.c_void_p), ("n_segments", C.c_int64),
                ("xyz", C.c_void_p), ("sdf_gt", C.c_void_p), ("n_points", C.c_int64), ("n_norm", C.c_int64),
                ("row_offset", C.c_int64), ("seg_len", C.c_int64)]


class DsdfLossCfg(C.Structure):
    _fields_ = [("clamp_dist", C.c_float), ("reg_coef", C.c_float), ("code_bound", C.c_float),
                ("training", C.c_int32), ("frozen_decoder", C.c_int32), ("dropout_key", C.c_uint32 * MAX_LAYERS),
                ("dw_phase", C.c_int32), ("dw_buckets", C.c_int32)]


class DsdfAdamCfg(C.Structure):
    _fields_ = [("step", C.c_int64), ("lr_decoder", C.c_float), ("lr_latent", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("grad_scale", C.c_void_p)]


PROF_CLASSES = 8
PROF_NAMES = ("gemm_nt_kernel", "gemm_tn_kernel", "last_layer_kernel", "fused_forward_kernel", "fused_backward_kernel",
              "dw_stream_kernel", "other", "fused_fwd_bwd_kernel")


class DsdfProfile(C.Structure):
    _fields_ = [("ms", C.c_double * PROF_CLASSES), ("flops", C.c_double * PROF_CLASSES),
                ("count", C.c_int64 * PROF_CLASSES), ("dropped", C.c_int32)]


class DsdfError(RuntimeError):
    pass


_P, _I64, _I32, _F, _SZ = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_size_t
_NET = C.POINTER(DsdfNet)

# name -> argtypes; every function returns int except dsdf_last_error
PROTOTYPES = {
    "dsdf_abi_version": [],
    "dsdf_param_layout": [_NET, C.POINTER(DsdfParamLayout)],
    "dsdf_packed_floats": [_NET, C.POINTER(_I64)],
    "dsdf_workspace_bytes": [_NET, _I64, _I64, C.POINTER(_SZ)],
    "dsdf_decode_workspace_bytes": [_NET, _I64, C.POINTER(_SZ)],
    "dsdf_workspace_bytes_buckets": [_NET, _I64, _I64, _I32, C.POINTER(_SZ)],
    "dsdf_dw_phase_supported": [_NET],
    "dsdf_grad_buckets": [_NET, _I32, C.POINTER(_I32), C.POINTER(_I64)],
    "dsdf_materialize_weights": [_NET, _P, _P, _P],
    "dsdf_decode": [_NET, _P, _P, _P, _I64, _I64, _P, _P, _SZ, _P],
    "dsdf_module_forward": [_NET, _P, _P, _P, _I64, _I64, _I32, C.POINTER(C.c_uint32), _P, _P, _SZ, _P],
    "dsdf_module_backward": [_NET, _P, _P, _P, _I64, _I32, C.POINTER(C.c_uint32), _P, _I32, _P, _I64, _P, _SZ, _P],
    "dsdf_module_jvp": [_NET, _P, _P, _P, _I64, _I64, _I32, C.POINTER(C.c_uint32), _P, _P, _SZ, _P],
    "dsdf_train_forward_backward": [_NET, _P, _P, _P, _I64, C.POINTER(DsdfBatch), C.POINTER(DsdfLossCfg), _P, _P, _P,
                                    _P, _I32, _P, _SZ, _P],
    "dsdf_grad_norm": [_P, _I64, _F, _P, _P, _P, _SZ, _P],
    "dsdf_adam_step": [_NET, _P, _P, _P, _P, _P, _P, _P, _P, _I64, C.POINTER(DsdfAdamCfg), _P, _P],
    "dsdf_train_step": [_NET, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, C.POINTER(DsdfBatch), C.POINTER(DsdfLossCfg),
                        C.POINTER(DsdfAdamCfg), _P, _P, _P, _SZ, _P],
    "dsdf_adam_latent_only": [_P, _P, _P, _P, _I64, C.POINTER(DsdfAdamCfg), _P],
    "dsdf_adam_latent_sched": [_P, _P, _P, _P, _I64, _P, _I64, _P, _F, _F, _F, _F, _P],
    "dsdf_profile_enable": [_I32],
    "dsdf_profile_read": [C.POINTER(DsdfProfile)],
    "dsdf_gemm_nt": [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P, _P],
    "dsdf_gemm_tn": [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P, _SZ, _P],
    "dsdf_dropout_mask": [C.c_uint32, _F, _I64, _I64, _I64, _P, _P],
    "dsdf_decode_latent": [_NET, _P, _P, _P, _P, _I64, _P, _P, _SZ, _P],
    "dsdf_decode_latent_supported": [_NET],
    "dsdf_sample_batch": [_P, C.c_int32, _P, _P, _P, _P, _P, _I64, _I64, C.c_uint64, _P, _P, _P],
    "dsdf_sample_batch_seq": [_P, C.c_int32, _P, _P, _P, _P, _P, _I64, _I64, C.c_uint64, C.c_uint64, _P, _P, _P, _P],
}

_lib = None


def lib():
    """Load libdsdf_hip.so once.  Raises if it has not been built (python -m deepsdf_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64; it must be in the process BEFORE our library is dlopen'ed so that both use
    # ONE HIP runtime (loaded the other way round, ours binds /opt/rocm's copy and sees "no ROCm-capable device")
    import torch  # noqa: F401
    if not os.path.exists(LIB):
        raise DsdfError(f"{LIB} is missing: build it with `python -m deepsdf_amd.build` "
                        "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    h = C.CDLL(LIB)
    for name, args in PROTOTYPES.items():
        fn = getattr(h, name)  # AttributeError here = header/library mismatch
        fn.argtypes = args
        fn.restype = C.c_int
    h.dsdf_last_error.argtypes = []
    h.dsdf_last_error.restype = C.c_char_p
    v = h.dsdf_abi_version()
    if v != ABI_VERSION:
        raise DsdfError(f"libdsdf_hip.so ABI {v} != binding ABI {ABI_VERSION}: rebuild")
    _lib = h
    return h


def check(rc):
    if rc != 0:
        raise DsdfError(f"libdsdf_hip error {rc}: {lib().dsdf_last_error().decode()}")
