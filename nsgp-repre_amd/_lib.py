"""ctypes binding of libnsgp_repre_hip.so (the C ABI declared in include/nsgp_repre.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libnsgp_repre_hip.so"
_lib = None

NSGP_OPT_SGD, NSGP_OPT_ADAM = 0, 1
NSGP_MAX_HYPER = 32


class TensorDesc(C.Structure):
    """nsgp_tensor_t"""
    _fields_ = [("param", C.c_void_p), ("state0", C.c_void_p), ("state1", C.c_void_p), ("state2", C.c_void_p),
                ("proj", C.c_void_p), ("numel", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32),
                ("hyper", C.c_int32), ("rank", C.c_int32), ("basis", C.c_void_p), ("basis_scale", C.c_float),
                ("split_kind", C.c_int32), ("proj_split", C.c_void_p), ("split_scale", C.c_float), ("reserved", C.c_int32),
                ("basis_rows", C.c_void_p)]


class Hyper(C.Structure):
    """nsgp_hyper_t"""
    _fields_ = [("lr", C.c_float), ("momentum", C.c_float), ("one_minus_dampening", C.c_float),
                ("weight_decay", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("one_minus_beta1", C.c_float), ("one_minus_beta2", C.c_float), ("eps", C.c_float),
                ("step_size", C.c_float), ("decoupled_decay", C.c_float), ("nesterov", C.c_int32),
                ("first_step", C.c_int32), ("amsgrad", C.c_int32), ("write_grad", C.c_int32),
                ("reserved", C.c_int32)]


class CovGeom(C.Structure):
    """nsgp_cov_geom_t"""
    _fields_ = [(k, C.c_int32) for k in ("batch", "cin", "h", "w", "kh", "kw", "sh", "sw", "ph", "pw")]


# every exported symbol of include/nsgp_repre.h: name -> (restype, argtypes)
SIGNATURES = {
    "nsgp_abi_version": (C.c_int, []),
    "nsgp_last_error": (C.c_char_p, []),
    "nsgp_device_count": (C.c_int, []),
    "nsgp_device_arch": (C.c_int, [C.c_char_p, C.c_int]),
    "nsgp_debug_install_abort_backtrace": (C.c_int, []),
    "nsgp_plan_workspace_bytes": (C.c_size_t, [C.POINTER(TensorDesc), C.c_int, C.c_int]),
    "nsgp_plan_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(TensorDesc), C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "nsgp_plan_destroy": (C.c_int, [C.c_void_p]),
    "nsgp_plan_step": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(Hyper), C.c_int, C.c_void_p]),
    "nsgp_plan_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nsgp_plan_launch_shape": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "nsgp_plan_profile_detail": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "nsgp_plan_lowrank_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nsgp_plan_profile_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "nsgp_plan_profile_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "nsgp_project": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "nsgp_cov_workspace_bytes": (C.c_size_t, [C.c_int] * 9),
    "nsgp_cov_accumulate_conv2d": (C.c_int, [C.c_void_p] + [C.c_int] * 10 + [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "nsgp_cov_plan_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(CovGeom), C.c_int]),
    "nsgp_cov_plan_destroy": (C.c_int, [C.c_void_p]),
    "nsgp_cov_plan_workspace_bytes": (C.c_size_t, [C.c_void_p]),
    "nsgp_cov_plan_routes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    "nsgp_cov_plan_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "nsgp_cov_plan_forms": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "nsgp_cov_set_corr_mode": (C.c_int, [C.c_int]),
    "nsgp_cov_plan_run": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p, C.c_size_t, C.c_void_p]),
    "nsgp_cov_accumulate_linear": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "nsgp_projector_scratch_bytes": (C.c_size_t, [C.c_int]),
    "nsgp_build_projector": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "nsgp_build_projector_head": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "nsgp_ewc_loss": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nsgp_ewc_grad": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "repre_pseudo_label_filter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float,
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "repre_nms_workspace_bytes": (C.c_size_t, [C.c_int]),
    "repre_nms": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "repre_replay_ce_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "repre_replay_ce_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "repre_replay_head_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "repre_replay_head_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                            C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p]),
    "repre_replay_head_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                             C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p,
                                                                                            C.c_size_t, C.c_void_p]),
    "nsgp_plan_uses_split_mfma": (C.c_int, [C.c_void_p]),
    "nsgp_plan_tile_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nsgp_cov_set_split_mfma": (C.c_int, [C.c_int]),
    "nsgp_split_projector_f16_bytes": (C.c_size_t, [C.c_int]),
    "nsgp_split_projector_f16": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "repre_sim_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "repre_sim_counts": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_void_p]),
    "repre_masked_mean_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "repre_masked_mean": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
}


def lib_path():
    return os.path.join(_HERE, _LIB_NAME)


def load_library():
    """Load the HIP library or raise -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{_LIB_NAME} not found at {path}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  nsgp_repre_amd has no CPU/eager fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load_library().nsgp_last_error()
        raise RuntimeError(f"{what or 'nsgp call'} failed with code {rc}: {msg.decode() if msg else ''}")
