"""ctypes binding of libmasklab_hip.so (the C ABI declared in include/masklab_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a kernel
call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The ONE library the product path loads: the in-tree build.  No environment override -- an experiment that wants another
# build assigns `_lib.LIB_PATH` in its own script before the first load() (scripts/h256_ab.py); bench.py prints the
# resolved path into its JSON line.
LIB_PATH = os.path.join(_HERE, "libmasklab_hip.so")

ABI_VERSION = 7          # ML_ABI_VERSION of include/masklab_hip.h
ACT_NONE, ACT_RELU, ACT_RELU6, ACT_SIGMOID = 0, 1, 2, 3
ACT_BY_NAME = {None: ACT_NONE, "linear": ACT_NONE, "relu": ACT_RELU, "relu6": ACT_RELU6,
               "sigmoid": ACT_SIGMOID}


class ConvDesc(C.Structure):
    """Mirror of `ml_conv2d_desc` (include/masklab_hip.h)."""
    _fields_ = [
        ("in_", C.c_void_p), ("wgt", C.c_void_p), ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("out", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("in_cstride", C.c_int32), ("in_coff", C.c_int32),
        ("span", C.c_int32), ("span_pad", C.c_int32), ("cpp_shift", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("dil", C.c_int32),
        ("pad_t", C.c_int32), ("pad_l", C.c_int32),
        ("cout", C.c_int32), ("n_pad", C.c_int32),
        ("out_cstride", C.c_int32), ("out_coff", C.c_int32),
        ("res_cstride", C.c_int32), ("res_coff", C.c_int32),
        ("act", C.c_int32), ("group_cin_step", C.c_int32), ("shuffle2x2", C.c_int32),
        ("tile", C.c_int32), ("math", C.c_int32), ("out_f16", C.c_int32),
        ("out_bstride", C.c_int64),
        ("live", C.c_void_p), ("live_period", C.c_int32), ("reserved1", C.c_int32),
        ("gn_partials", C.c_void_p),
    ]


class GnDesc(C.Structure):
    """Mirror of `ml_gn_desc` (include/masklab_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("HWC", C.c_int64), ("N", C.c_int32), ("C", C.c_int32), ("G", C.c_int32), ("relu", C.c_int32),
                ("out_cstride", C.c_int32), ("out_coff", C.c_int32), ("eps", C.c_float), ("dtype", C.c_int32),
                ("live", C.c_void_p), ("live_period", C.c_int32), ("reserved", C.c_int32),
                ("partials", C.c_void_p), ("n_partials", C.c_int32), ("reserved2", C.c_int32)]


class DeconvOutProblem(C.Structure):
    """Mirror of `ml_deconv_out_problem` (include/masklab_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("wd", C.c_void_p), ("bd", C.c_void_p), ("wo_table", C.c_void_p),
                ("bo", C.c_void_p), ("out", C.c_void_p), ("M", C.c_int64), ("hw", C.c_int32), ("w", C.c_int32),
                ("rois_per_image", C.c_int32), ("reserved0", C.c_int32), ("out_image_stride", C.c_int64),
                ("out_base", C.c_int64), ("live", C.c_void_p)]


GN_MAX_PROBLEMS = 8
DECONV_OUT_MAX_PROBLEMS = 4
_i32, _i64, _f32, _vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p

# name -> (restype, argtypes); every symbol include/masklab_hip.h declares
SIGNATURES = {
    "ml_version": (C.c_int, []),
    "ml_last_error": (C.c_char_p, []),
    "ml_device_check": (C.c_int, []),
    "ml_conv2d_f32": (C.c_int, [C.POINTER(ConvDesc), _vp]),
    "ml_conv2d_ntile": (C.c_int, [_i32, _i32]),
    "ml_conv2d_uses_pipe": (C.c_int, [C.POINTER(ConvDesc)]),
    "ml_conv2d_launch_ntile": (C.c_int, [C.POINTER(ConvDesc), _i32, _i32]),
    "ml_conv2d_launch_mtile": (C.c_int, [C.POINTER(ConvDesc), _i32, _i32]),
    "ml_conv2d_launch_splits": (C.c_int, [C.POINTER(ConvDesc), _i32, _i64, C.POINTER(C.c_int32)]),
    "ml_conv2d_gn_min_launch_tiles": (_i64, []),
    "ml_conv2d_workspace_bytes": (_i64, []),
    "ml_conv2d_multi_f32": (C.c_int, [C.POINTER(ConvDesc), _i32, _vp, _i64, _vp]),
    "ml_deconv2x2_out1x1_f32": (C.c_int, [C.POINTER(DeconvOutProblem), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ml_gconv3x3_f32": (C.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 11 + [_vp]),
    "ml_gconv3x3_f16": (C.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 11 + [_vp]),
    "ml_maxpool3x3s2_f16": (C.c_int, [_vp, _vp] + [_i32] * 8 + [_vp]),
    "ml_stem7x7s2_pool_f16": (C.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 5 + [_vp]),
    "ml_stem7x7s2_pool_f32": (C.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 5 + [_vp]),
    "ml_stem7x7s2_pool_x3": (C.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 5 + [_vp]),
    "ml_subsample2_f16": (C.c_int, [_vp, _vp] + [_i32] * 4 + [_vp]),
    "ml_cast_f16_to_f32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "ml_cast_f32_to_f16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "ml_resize_bilinear_ac_f16": (C.c_int, [_vp, _vp, _vp] + [_i32] * 12 + [_vp]),
    "ml_dwconv3x3_f16": (C.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 15 + [_vp]),
    "ml_global_mean_f16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "ml_groupnorm_chunk_f16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _f32, _i32, _i32, _i32, _vp, _vp]),
    "ml_roi_crop_resize_f16": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp] + [_i32] * 10 + [_f32, _f32, _i32, _i32, _vp, _vp]),
    "ml_deconv2x2_out1x1_f16": (C.c_int, [C.POINTER(DeconvOutProblem), _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ml_dwconv3x3_f32": (C.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 15 + [_vp]),
    "ml_maxpool3x3s2_f32": (C.c_int, [_vp, _vp] + [_i32] * 8 + [_vp]),
    "ml_preprocess_f32": (C.c_int, [_vp, _i32, _vp, _i64, _i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                    C.POINTER(C.c_float), _vp]),
    "ml_groupnorm_workspace_bytes": (_i64, [_i32, _i32]),
    "ml_groupnorm_multi_f32": (C.c_int, [C.POINTER(GnDesc), _i32, _vp, _i64, _vp]),
    "ml_groupnorm_chunk_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _f32, _i32, _i32, _i32, _vp, _vp]),
    "ml_resize_bilinear_ac_f32": (C.c_int, [_vp, _vp, _vp] + [_i32] * 12 + [_vp]),
    "ml_global_mean_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "ml_scale_channels_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "ml_restore_boxes_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "ml_detection_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "ml_detection_proposal_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _f32, _i32, _vp, _vp]),
    "ml_mask_distribute_i32": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "ml_roi_crop_resize_f32": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp] + [_i32] * 10 + [_f32, _f32, _i32, _i32, _vp, _vp]),
    "ml_mold_levels_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, C.POINTER(C.c_int32), _vp]),
    "ml_mold_levels_dev_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp]),
    "ml_add_f32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "ml_fill_f32": (C.c_int, [_vp, _f32, _i64, _vp]),
    "ml_resize_image_ac": (C.c_int, [_vp, _i32, _vp, _vp, _f32] + [_i32] * 6 + [_vp]),
    "ml_trim_instances_f32": (C.c_int, [_vp] * 5 + [_i32] * 5 + [_vp]),
    "ml_upsample_boxes_i32": (C.c_int, [_vp, _vp, _i64, _f32, _f32, _vp]),
    "ml_threshold_i32": (C.c_int, [_vp, _vp, _f32, _i64, _vp]),
    "ml_semantic_smoothing_f32": (C.c_int, [_vp, _vp, _vp] + [_i32] * 4 + [_vp, _vp, _vp]),
    "ml_crop_pad_mask_f32": (C.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 6 + [_vp]),
    "ml_nonzero_bbox_i32": (C.c_int, [_vp] + [_i32] * 5 + [_vp, _vp]),
    "ml_instance_summary_workspace_bytes": (_i64, [_i32, _i32]),
    "ml_instance_summary_f32": (C.c_int, [_vp, _i32, _i32, _vp, _vp] + [_i32] * 4 + [_f32, _f32, _vp, _vp]),
    "ml_instance_summary_rois_f32": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp] + [_i32] * 6 + [_f32, _f32, _vp, _vp]),
}

_lib = None


def load():
    """Load the shared library (once) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"masklab_hip: {LIB_PATH} is missing -- build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            f"There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.ml_version() != ABI_VERSION:       # e.g. a stale .so from before ml_conv2d_desc changed layout
        raise RuntimeError(f"masklab_hip: {LIB_PATH} has ABI version {lib.ml_version()}, this package binds "
                           f"version {ABI_VERSION}: rebuild the library")
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().ml_last_error()
        raise RuntimeError(f"masklab_hip: {what} failed with status {status}: "
                           f"{msg.decode() if msg else '?'}")
