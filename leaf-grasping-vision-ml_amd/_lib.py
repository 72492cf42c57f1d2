"""ctypes binding of liblgrasp.so (include/leafgrasp.h).  The library is REQUIRED: there is no CPU or
PyTorch fallback anywhere in this package -- a missing/unloadable library raises at import."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LG_LIB_PATH") or os.path.join(_HERE, "liblgrasp.so")  # override: kernel experiments only

LG_NUM_MAPS = 8
MAP_NAMES = ("sdf_score", "approach_score", "flatness_map", "isolation_map", "distance_map",
             "accessibility_map", "stem_penalty", "traditional_score")  # grasp_point_selector.py:258-280
MAP_INDEX = {n: i for i, n in enumerate(MAP_NAMES)}

LG_OK, LG_ERR_INVALID, LG_ERR_HIP, LG_ERR_NOMEM, LG_ERR_NO_MODEL, LG_ERR_UNSUPPORTED, LG_ERR_BUSY = 0, -1, -2, -3, -4, -5, -6


class LgParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("cx", "cy", "f")] + [(n, C.c_float) for n in (
        "w_approach", "w_sdf", "w_flat", "w_access", "sdf_w_interior", "sdf_w_align",
        "sdf_w_sdf", "optimal_distance", "access_w_dist", "access_w_dir", "flat_scale", "iso_w_close",
        "iso_w_wide", "iso_ramp_top", "iso_ramp_bottom", "min_edge_distance", "stem_valid_thresh")] + \
        [(n, C.c_int32) for n in ("stem_se", "stem_bottom_div", "top_k", "nms_min_distance",
                                  "pregrasp_clearance", "mask_is_bool", "gaussian_size", "chamfer_init_dist0", "reserved_")]


_FP = C.POINTER(C.c_float)


class LgCnnWeights(C.Structure):
    _fields_ = [("conv_w", _FP * 8), ("conv_b", _FP * 8), ("bn_g", _FP * 8), ("bn_b", _FP * 8),
                ("bn_m", _FP * 8), ("bn_v", _FP * 8), ("att_w", _FP), ("att_b", _FP),
                ("fc_w", _FP * 4), ("fc_b", _FP * 4), ("fbn_g", _FP * 3), ("fbn_b", _FP * 3),
                ("fbn_m", _FP * 3), ("fbn_v", _FP * 3), ("bn_eps", C.c_float), ("attention_type", C.c_int32),
                ("ca_w1", _FP), ("ca_b1", _FP), ("ca_w2", _FP), ("ca_b2", _FP),
                ("n_blocks", C.c_int32), ("filters", C.c_int32 * 4)]


class LgGraspResult(C.Structure):
    _fields_ = [("found", C.c_int32), ("x", C.c_int32), ("y", C.c_int32), ("X", C.c_float), ("Y", C.c_float),
                ("Z", C.c_float), ("has_pre", C.c_int32), ("pX", C.c_float), ("pY", C.c_float),
                ("pZ", C.c_float), ("n_candidates", C.c_int32), ("ml_used", C.c_int32),
                ("best_score", C.c_float), ("theta", C.c_float)]


class LgTrainHparams(C.Structure):
    """lg_train_hparams; defaults = scripts/train_model.py:221-222,256."""
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("max_grad_norm", C.c_float), ("pos_weight", C.c_float)]


class LgLeafStat(C.Structure):
    _fields_ = [("id", C.c_int32), ("area", C.c_int32), ("touches_border", C.c_int32), ("pad_", C.c_int32),
                ("sum_x", C.c_double), ("sum_y", C.c_double), ("sum_depth", C.c_double), ("sum_ray", C.c_double),
                ("median_depth", C.c_float), ("pad2_", C.c_float)]


# every symbol include/leafgrasp.h declares: (restype, argtypes)
_VP = C.c_void_p
SYMBOLS = {
    "lg_create": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "lg_destroy": (C.c_int, [_VP]),
    "lg_last_error": (C.c_char_p, [_VP]),
    "lg_orientation_note": (C.c_char_p, [_VP]),
    "lg_version": (C.c_char_p, []),
    "lg_default_params": (None, [C.POINTER(LgParams)]),
    "lg_score_maps": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.POINTER(LgParams),
                                C.POINTER(_VP * LG_NUM_MAPS), _VP, _FP, _VP]),
    "lg_smooth_depth": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, _VP, _VP]),
    "lg_gaussian_taps": (C.c_int, [C.c_int, _FP]),
    "lg_topk_nms": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _VP, _VP, _VP]),
    "lg_gather_patches": (C.c_int, [_VP, _VP, _VP, C.POINTER(_VP * LG_NUM_MAPS), C.c_int, C.c_int, C.c_int,
                                    C.c_int, _VP, _VP, _VP, _VP]),
    "lg_cnn_load": (C.c_int, [_VP, C.POINTER(LgCnnWeights)]),
    "lg_cnn_unload": (C.c_int, [_VP]),
    "lg_cnn_forward": (C.c_int, [_VP, _VP, C.c_int, _VP, _VP]),
    "lg_select_grasp": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.POINTER(LgParams),
                                  C.POINTER(_VP * LG_NUM_MAPS), _VP, C.POINTER(LgGraspResult), _VP]),
    "lg_select_grasp_labels": (C.c_int, [_VP, _VP, _VP, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.POINTER(LgParams),
                                         C.POINTER(_VP * LG_NUM_MAPS), _VP, C.POINTER(LgGraspResult), _VP]),
    "lg_leaf_stats": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                C.POINTER(LgLeafStat), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int32), _VP]),
    "lg_leaf_stats_batch": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                      C.POINTER(LgLeafStat), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int), _VP]),
    "lg_leaf_select_batch": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, _VP]),
    "lg_leaf_select_from_stats": (C.c_int, [C.POINTER(LgLeafStat), C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_double,
                                            C.c_double, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int,
                                            C.POINTER(C.c_int32)]),
    "lg_leaf_orientation": (C.c_int, [_VP, _VP, C.c_int, C.c_int, _FP, C.POINTER(C.c_int), _VP]),
    "lg_profile_enable": (C.c_int, [_VP, C.c_int]),
    "lg_profile_read": (C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "lg_debug_dt_max": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]),
    "lg_format_grasp_results": (C.c_int, [_VP, C.c_int, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]),
    "lg_debug_dt_form": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_int32)]),
    "lg_harvest_patches": (C.c_int, [_VP, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "lg_negative_masks": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, _VP, _VP, _VP]),
    "lg_leaf_contour": (C.c_int, [_VP, _VP, C.c_int, C.c_int, _VP, C.c_int, C.POINTER(C.c_int), _VP]),
    "lg_train_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(_VP)]),
    "lg_train_destroy": (C.c_int, [_VP]),
    "lg_train_last_error": (C.c_char_p, [_VP]),
    "lg_train_sizes": (C.c_int, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "lg_train_set_state": (C.c_int, [_VP, _FP, _FP, _FP, _FP, C.c_int64]),
    "lg_train_get_state": (C.c_int, [_VP, _FP, _FP, _FP, _FP, _FP, C.POINTER(C.c_int64)]),
    "lg_train_step": (C.c_int, [_VP, _VP, _VP, C.c_int, _VP, C.c_uint64, C.POINTER(LgTrainHparams), C.c_int, _FP, _FP, _VP]),
    "lg_train_sync": (C.c_int, [_VP]),
    "lg_train_grad_buffer": (C.c_int, [_VP, C.POINTER(_FP), C.POINTER(C.c_int64)]),
    "lg_train_apply": (C.c_int, [_VP, C.POINTER(LgTrainHparams), _FP]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
    # PyTorch-ROCm ships its own libamdhip64 and every device pointer this package passes comes from torch: torch must be
    # the one that brings the HIP runtime into the process.  Loaded the other way round, liblgrasp.so binds the system
    # runtime in /opt/rocm, torch then loads its bundled copy, and the second runtime finds no device
    # ("lg_create: hipGetDeviceCount -> no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class LgError(RuntimeError):
    pass


def check(handle, rc, what):
    if rc == LG_ERR_BUSY:
        raise LgError(f"{what}: another thread is inside a call on this handle (one call in flight per selector instance)")
    if rc != LG_OK:
        msg = lib.lg_last_error(handle if handle else None).decode()   # NULL handle: the reason lg_create failed
        raise LgError(f"{what} failed with status {rc}: {msg}")


def default_params():
    p = LgParams()
    lib.lg_default_params(C.byref(p))
    return p
