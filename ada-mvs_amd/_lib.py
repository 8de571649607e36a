"""ctypes binding of libadamvs_hip.so (C ABI: include/adamvs_hip.h).

The library is loaded AFTER `import torch` so that its libamdhip64 dependency
binds to the HIP runtime torch already mapped (one runtime per process).
There is no fallback: if the library is missing or a call fails this raises.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported before the library is loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADAMVS_LIB_PATH") or os.path.join(_HERE, "libadamvs_hip.so")     # override: A/B of two builds

c_f = ctypes.c_void_p          # device pointer to float
c_i = ctypes.c_int
c_sz = ctypes.c_size_t
c_st = ctypes.c_void_p         # hipStream_t


class FuseWeights(ctypes.Structure):
    """adamvs_fuse_weights"""
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "conv1", "gates1", "gates1_b", "cand1", "cand1_b", "conv2", "gates2", "gates2_b",
        "cand2", "cand2_b", "upconv1", "upconv1_b", "final_w", "gates1_w", "gates2_w", "cand2_w", "cand1_w")]


class FConvWeights(ctypes.Structure):
    """adamvs_fconv_weights"""
    _fields_ = [("w", ctypes.c_void_p), ("b", ctypes.c_void_p)]


class ContextWeights(ctypes.Structure):
    """adamvs_context_weights"""
    _fields_ = [("w1", ctypes.c_void_p), ("b1", ctypes.c_void_p), ("w2", ctypes.c_void_p)]


class FeatureWeights(ctypes.Structure):
    """adamvs_feature_weights"""
    _fields_ = [(n, FConvWeights) for n in (
        "conv0_0", "conv0_1", "conv1_0", "conv1_1", "conv1_2", "conv2_0", "conv2_1", "conv2_2",
        "out1", "deconv1_t", "deconv1_c", "out2", "deconv2_t", "deconv2_c", "out3")] + \
               [(n, ContextWeights) for n in ("br1_1", "br1_2", "br2_1", "br2_2", "br3_1", "br3_2")]


class FeatureFpnWeights(ctypes.Structure):
    """adamvs_feature_fpn_weights"""
    _fields_ = [(n, FConvWeights) for n in (
        "conv0_0", "conv0_1", "conv1_0", "conv1_1", "conv1_2", "conv2_0", "conv2_1", "conv2_2",
        "out1", "inner1", "out2", "inner2", "out3")]


class StageDesc(ctypes.Structure):
    """adamvs_stage_desc"""
    _fields_ = [(n, ctypes.c_int) for n in ("B", "S", "C", "h", "w", "D", "in_up", "first_stage", "prev_h", "prev_w", "precision", "precision_fuse",
                                            "eps_in_numerator", "plane_mode")] + [("half_span", ctypes.c_float),
                                                                                   ("half_span_dev", ctypes.c_void_p)]


# name -> (restype, argtypes); every symbol include/adamvs_hip.h declares
SIGNATURES = {
    "adamvs_version": (c_i, []),
    "adamvs_last_error_string": (ctypes.c_char_p, []),
    "adamvs_option_count": (c_i, []),
    "adamvs_option_name": (ctypes.c_char_p, [c_i]),
    "adamvs_option_default": (c_i, [ctypes.c_char_p, ctypes.POINTER(c_i)]),
    "adamvs_get_option": (c_i, [ctypes.c_char_p, ctypes.POINTER(c_i)]),
    "adamvs_set_option": (c_i, [ctypes.c_char_p, c_i]),
    "adamvs_relative_transforms": (c_i, [c_f, c_f, c_i, c_i, c_st]),
    "adamvs_pack_features": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_st]),
    "adamvs_unpack_features": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_st]),
    "adamvs_depth_range_samples_uniform": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_st]),
    "adamvs_depth_range_samples_window": (c_i, [c_f, ctypes.c_double, c_f, c_i, c_i, c_i, c_i, c_st]),
    "adamvs_resize_bilinear": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_st]),
    "adamvs_depth_regression": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    "adamvs_homo_warp": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_st]),
    "adamvs_pair_similarity": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    "adamvs_cost_reg_net_2d_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "adamvs_cost_reg_width": (c_i, [c_i, c_i]),
    "adamvs_cost_reg_net_2d_weight_floats": (c_sz, [c_i, c_i]),
    "adamvs_cost_reg_net_2d": (c_i, [c_f, c_f, c_sz, c_f, c_i, c_i, c_i, c_i, c_i, ctypes.c_void_p, c_sz, c_st]),
    "adamvs_conv3x3_dd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    "adamvs_conv3x3_dd_wino": (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_st]),
    "adamvs_softmax_max_regress": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_st]),
    "adamvs_prob_softmax_regress": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    "adamvs_prob_softmax_regress_wino_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "adamvs_prob_softmax_regress_wino": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, ctypes.c_void_p, c_sz, c_st]),
    "adamvs_aggregate_conv1_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "adamvs_aggregate_conv1": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, ctypes.c_void_p, c_sz, c_st]),
    "adamvs_slice_reg_step_scratch_bytes": (c_sz, [c_i, c_i, c_i]),
    "adamvs_slice_reg_step": (c_i, [c_f, c_f, c_f, ctypes.POINTER(FuseWeights), c_f, c_i, c_i, c_i, c_i, c_i, c_i,
                                    ctypes.c_void_p, c_sz, c_st]),
    "adamvs_depth_stage_workspace_bytes": (c_sz, [ctypes.POINTER(StageDesc)]),
    "adamvs_recurrence_schedule": (c_i, [c_i, ctypes.c_longlong]),
    "adamvs_gru_wino_mask": (c_i, []),
    "adamvs_depth_stage_forward": (c_i, [ctypes.POINTER(StageDesc), c_f, c_f, c_f, c_f, c_f, c_sz, ctypes.POINTER(FuseWeights),
                                         c_f, c_f, c_f, c_f, c_i, ctypes.c_void_p, c_sz, c_st]),
    "adamvs_bench_stage_phase": (c_i, [ctypes.POINTER(StageDesc), c_f, c_f, c_f, c_f, c_f, c_sz, ctypes.POINTER(FuseWeights),
                                       c_f, c_f, c_f, c_f, c_i, ctypes.c_void_p, c_sz, c_st]),
    "adamvs_feature_net0_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "adamvs_feature_net0": (c_i, [c_f, ctypes.POINTER(FeatureWeights), c_f, c_f, c_f, c_i, c_i, c_i, ctypes.c_void_p, c_sz, c_st]),
    "adamvs_feature_net0_views": (c_i, [c_f, ctypes.POINTER(FeatureWeights), c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, ctypes.c_void_p, c_sz, c_st]),
    "adamvs_feature_net_fpn_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "adamvs_feature_net_fpn": (c_i, [c_f, ctypes.POINTER(FeatureFpnWeights), c_f, c_f, c_f, c_i, c_i, c_i, ctypes.c_void_p, c_sz, c_st]),
    "adamvs_red_variance_cost": (c_i, [c_f, c_f, c_f, c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_st]),
    "adamvs_channel_copy": (c_i, [c_f, c_f, c_i, c_i, c_i, ctypes.c_long, c_i, c_i, ctypes.c_long, c_i, c_i, c_st]),
    "adamvs_group_stats_workspace_bytes": (c_sz, [c_i, c_i]),
    "adamvs_group_stats_partial": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, ctypes.c_void_p, c_sz, c_st]),
    "adamvs_group_stats_finish": (c_i, [ctypes.c_void_p, c_f, c_i, c_i, c_i, c_i, ctypes.c_float, c_st]),
    "adamvs_gru2_gates_apply": (c_i, [c_f, c_f, c_i, ctypes.c_void_p, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, ctypes.c_float, c_st]),
    "adamvs_conv3x3_pair": (c_i, [c_f, c_i, c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_st]),
    "adamvs_gru2_out_apply": (c_i, [c_f, ctypes.c_void_p, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, ctypes.c_float, c_st]),
    "adamvs_red_recur_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "adamvs_red_recur_pair": (c_i, [c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, ctypes.c_float,
                                    ctypes.c_void_p, c_sz, c_st]),
    "adamvs_red_recur_split": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, ctypes.c_float,
                                     ctypes.c_void_p, c_sz, c_st]),
    "adamvs_soft_argmin": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_st]),
}

ABI_VERSION = 16
PRECISIONS = {"fp32": 0, "bf16x3": 1}
PLANES_EXPLICIT, PLANES_UNIFORM, PLANES_WINDOW = 0, 1, 2
PHASE_VIEW_WEIGHTS, PHASE_AGGREGATE, PHASE_RECURRENCE, PHASE_SOFT_ARGMIN, PHASE_ALL = 1, 2, 4, 8, 15
_lib = None


class AdaMVSHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdaMVSHipError(
            "libadamvs_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "-- there is no CPU fallback for the Ada-MVS hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    v = lib.adamvs_version()
    if v != ABI_VERSION:
        raise AdaMVSHipError("libadamvs_hip.so ABI version %d, host expects %d: rebuild" % (v, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    """Map the C status code to an exception (0 ok, <0 argument error, >0 hipError_t)."""
    if rc == 0:
        return
    msg = load().adamvs_last_error_string().decode("utf-8", "replace")
    kind = "invalid argument" if rc < 0 else "HIP error %d" % rc
    raise AdaMVSHipError("%s failed (%s): %s" % (what, kind, msg))


def set_option(name, value):
    """adamvs_set_option: one of the integers of include/adamvs_hip.h "OPTIONS" (which of two equivalent kernel forms a layer takes)."""
    check(load().adamvs_set_option(name.encode(), int(value)), "adamvs_set_option(%s)" % name)


def get_option(name):
    v = c_i(0)
    check(load().adamvs_get_option(name.encode(), ctypes.byref(v)), "adamvs_get_option(%s)" % name)
    return v.value


def option_names():
    lib = load()
    return [lib.adamvs_option_name(i).decode() for i in range(lib.adamvs_option_count())]


class options:
    """`with _lib.options(winograd=0, gru_wino=0): ...` -- set for the block, restored afterwards (tests, A/B timing)."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.saved = {k: get_option(k) for k in self.kv}
        for k, v in self.kv.items():
            set_option(k, v)
        return self

    def __exit__(self, *a):
        for k, v in self.saved.items():
            set_option(k, v)
