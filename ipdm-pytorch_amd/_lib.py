"""ctypes binding of libipdm_hip.so (include/ipdm_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails, this module
raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C ipdm-pytorch_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IPDM_LIB_PATH") or os.path.join(_HERE, "libipdm_hip.so")    # (override: diagnostic builds)


class IpdmError(RuntimeError):
    pass


class FbpGeom(C.Structure):
    _fields_ = [("n_views", C.c_int32), ("n_det", C.c_int32), ("grid_n", C.c_int32),
                ("da", C.c_double), ("det_offset", C.c_double), ("dtheta_deg", C.c_double),
                ("source_origin", C.c_double), ("fov_half", C.c_double)]


class UnetCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("model_channels", C.c_int32), ("out_channels", C.c_int32),
                ("num_res_blocks", C.c_int32), ("num_heads", C.c_int32), ("n_mult", C.c_int32),
                ("n_attn", C.c_int32), ("channel_mult", C.c_double * 16),
                ("attention_resolutions", C.c_int32 * 16)]


_vp, _i32, _i64, _u64, _f32, _f64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); restype int => status code checked by _call
ABI_VERSION = 5          # ipdm_abi_version() of the header these prototypes were written against
PROF_CLASSES = 8         # kernel classes of ipdm_profile_end (include/ipdm_hip.h)

PROTOTYPES = {
    "ipdm_last_error": (C.c_char_p, []),
    "ipdm_abi_version": (C.c_int, []),
    "ipdm_set_option": (C.c_int, [C.c_char_p, _i32]),
    "ipdm_get_option": (C.c_int, [C.c_char_p, C.POINTER(_i32)]),
    "ipdm_fbp_plan_create": (C.c_int, [C.POINTER(FbpGeom), C.POINTER(_vp)]),
    "ipdm_fbp_plan_destroy": (C.c_int, [_vp]),
    "ipdm_fbp_workspace_bytes": (_sz, [_vp, _i32]),
    "ipdm_fbp_forward": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _f32, _vp, _sz, _vp]),
    "ipdm_fbp_filter": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _f32, _vp]),
    "ipdm_fbp_backproject": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "ipdm_fbp_index_map": (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "ipdm_fbp_table": (_i64, [_vp, _i32, _vp, _i64]),
    "ipdm_sharpen3x3": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "ipdm_schedule_create": (C.c_int, [_i32, _f64, C.POINTER(_vp)]),
    "ipdm_schedule_destroy": (C.c_int, [_vp]),
    "ipdm_schedule_coeffs": (C.c_int, [_vp, _i32, C.POINTER(_f32 * 8)]),
    "ipdm_schedule_alpha_cumprod": (C.c_int, [_vp, _i32, C.POINTER(_f32)]),
    "ipdm_cosine_lambda": (C.c_int, [_i32, _f64, _i32, C.POINTER(_f64)]),
    "ipdm_ddim_step": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _f64, _f64, _i32, _vp, _sz, _vp]),
    "ipdm_randn": (C.c_int, [_vp, _i32, _i64, _u64, _i64, _i64, _vp]),
    "ipdm_q_sample": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i64, _vp]),
    "ipdm_ddpm_workspace_bytes": (_sz, [_i32]),
    "ipdm_ddpm_step": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f64, _vp, _i32, _i32,
                                 _i32, _vp, _sz, _vp]),
    "ipdm_clamp": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "ipdm_axpbypcz": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f64, _f64, _f64, _vp]),
    "ipdm_guidance_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "ipdm_guidance_map": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f64, _i32, C.POINTER(_f64),
                                    C.POINTER(_f64), _vp, _sz, _vp]),
    "ipdm_lambda_ratio": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "ipdm_slice_median": (C.c_int, [_vp, _vp, _i32, _i64, _vp, _sz, _vp]),
    "ipdm_unet_param_count": (C.c_int, [C.POINTER(UnetCfg)]),
    "ipdm_unet_param_info": (C.c_int, [C.POINTER(UnetCfg), _i32, C.c_char_p, _i32, C.POINTER(_i32 * 4),
                                       C.POINTER(_i32)]),
    "ipdm_unet_create": (C.c_int, [C.POINTER(UnetCfg), C.POINTER(_vp), _i32, C.POINTER(_vp)]),
    "ipdm_unet_destroy": (C.c_int, [_vp]),
    "ipdm_unet_workspace_bytes": (_sz, [_vp, _i32, _i32, _i32]),
    "ipdm_unet_forward": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "ipdm_unet_forward_graph": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    "ipdm_op_conv2d": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32,
                                 _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ipdm_op_conv_gn_conv": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp,
                                       _i32, _vp, _vp, C.POINTER(_i32), _vp]),
    "ipdm_op_up_conv_chain": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp,
                                        _i32, _i32, _vp, _vp, C.POINTER(_i32), _vp]),
    "ipdm_profile_begin": (C.c_int, [_i32]),
    "ipdm_profile_begin_classes": (C.c_int, [_i32, C.c_uint32]),
    "ipdm_clock_probe": (C.c_int, [_vp, _i32, _i32, _vp]),
    "ipdm_profile_end": (C.c_int, [C.POINTER(_f64 * PROF_CLASSES), C.POINTER(_f64 * PROF_CLASSES), C.POINTER(_i64 * PROF_CLASSES), _i32]),
    "ipdm_bench_conv2d": (C.c_int, [_i32] * 11 + [C.POINTER(_f32)]),
    "ipdm_bench_attention": (C.c_int, [_i32] * 5 + [C.POINTER(_f32)]),
    "ipdm_op_attention": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "ipdm_conv_layout_code": (_i32, [_i32, _i32, _i32]),
    "ipdm_conv_kernel_code": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "ipdm_conv_kernel_code_stats": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "ipdm_attention_kernel_code": (_i32, [_i32]),
    "ipdm_art_plan_create": (C.c_int, [_vp, _vp, _vp, C.POINTER(_vp)]),
    "ipdm_art_plan_destroy": (C.c_int, [_vp]),
    "ipdm_art_workspace_bytes": (C.c_size_t, [_vp, _i32]),
    "ipdm_art_reconstruct": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, C.c_size_t, _vp]),
    "ipdm_art_project": (C.c_int, [_vp, _vp, _vp, _i32, _vp, C.c_size_t, _vp]),
}

_lib = None


def lib():
    """Loads libipdm_hip.so; raises IpdmError (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise IpdmError("%s not found: the HIP extension is mandatory (no CPU fallback). "
                            "Run __graft_entry__.build()." % LIB_PATH)
        # torch first: it ships its own libamdhip64, and the library must bind to the SAME runtime instance (loaded the other
        # way round -- e.g. build() followed by smoke() in one process -- the second runtime sees no device)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)       # AttributeError => header/library mismatch
            fn.restype = res
            fn.argtypes = args
        got = handle.ipdm_abi_version()
        if got != ABI_VERSION:
            raise IpdmError("%s has ABI version %d, this package binds version %d (include/ipdm_hip.h): rebuild with "
                            "__graft_entry__.build()" % (LIB_PATH, got, ABI_VERSION))
        _lib = handle
    return _lib


def call(name, *args):
    """Calls a status-returning entry point and raises IpdmError on a non-zero status."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise IpdmError("%s failed (%d): %s" % (name, rc, lib().ipdm_last_error().decode()))
    return rc


def set_option(name, value):
    """ipdm_set_option: process-wide library switch (README.md lists them); returns the previous value."""
    old = get_option(name)
    call("ipdm_set_option", name.encode(), int(value))
    return old


def get_option(name):
    v = C.c_int32()
    call("ipdm_get_option", name.encode(), C.byref(v))
    return v.value


class option:
    """with _lib.option("conv_no_wino", 1): ...  -- sets a library switch for the block and restores it afterwards."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def ptr(t):
    """Device (or host) address of a contiguous torch tensor / numpy array, as c_void_p."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        assert t.is_contiguous(), "non-contiguous tensor handed to the C ABI"
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
