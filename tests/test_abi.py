"""CPU-side checks of the drop-in boundary: libipdm_hip.so loads without a GPU, exports every symbol
include/ipdm_hip.h declares, the ctypes table in _lib.py covers exactly that set, and the host-only
entry points (no kernel launch) behave.  No compute call is made here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ipdm_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ipdm_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_path():
    names = _declared()
    for must in ("ipdm_fbp_forward", "ipdm_unet_forward", "ipdm_ddpm_step", "ipdm_q_sample", "ipdm_guidance_map",
                 "ipdm_lambda_ratio", "ipdm_sharpen3x3", "ipdm_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from ipdm_pytorch_amd import _lib
    assert os.path.isfile(_lib.LIB_PATH), "libipdm_hip.so missing: run __graft_entry__.build()"
    h = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(h, n)]
    assert not missing, "declared in include/ipdm_hip.h but not exported: %s" % missing


def test_ctypes_table_matches_header():
    from ipdm_pytorch_amd import _lib
    assert sorted(_lib.PROTOTYPES) == _declared()


def test_host_only_entry_points():
    """Schedule tables and UNet parameter inventory are host code: callable without a GPU."""
    from ipdm_pytorch_amd import _lib
    lib = _lib.lib()
    assert lib.ipdm_abi_version() == _lib.ABI_VERSION == 5
    out = C.c_double()
    _lib.call("ipdm_cosine_lambda", 15, 1.0, 3, C.byref(out))
    assert 0.0 <= out.value <= 0.999
    cfg = _lib.UnetCfg()
    cfg.in_channels, cfg.model_channels, cfg.out_channels, cfg.num_res_blocks, cfg.num_heads = 1, 64, 1, 2, 4
    mult = [1, 1, 2, 2, 4, 4]
    cfg.n_mult, cfg.n_attn = len(mult), 2
    for i, m in enumerate(mult):
        cfg.channel_mult[i] = m
    cfg.attention_resolutions[0], cfg.attention_resolutions[1] = 8, 16
    n = lib.ipdm_unet_param_count(C.byref(cfg))
    assert n > 0
    total = 0
    name = C.create_string_buffer(128)
    shape, nd = (C.c_int32 * 4)(), C.c_int32()
    for i in range(n):
        _lib.call("ipdm_unet_param_info", C.byref(cfg), i, name, 128, C.byref(shape), C.byref(nd))
        k = 1
        for d in range(nd.value):
            k *= shape[d]
        total += k
    assert total == 29_094_465 or abs(total - 29.09e6) < 0.01e6     # 29.09 M parameters (img net)


def test_errors_are_status_codes_not_exceptions():
    from ipdm_pytorch_amd import _lib
    lib = _lib.lib()
    rc = lib.ipdm_unet_param_count(None)
    assert rc < 0 and lib.ipdm_last_error()
    with pytest.raises(_lib.IpdmError):
        _lib.call("ipdm_cosine_lambda", 0, 1.0, 0, C.byref(C.c_double()))


def test_options_table_without_a_gpu():
    """ipdm_set_option / ipdm_get_option are host-only: defaults, the IPDM_ prefix alias, unknown names."""
    from ipdm_pytorch_amd import _lib
    assert _lib.get_option("conv_nm") == 0 and _lib.get_option("direct_max_cin") == 160 and _lib.get_option("unet_transpose") == -1
    with _lib.option("conv_no_up2", 1):
        assert _lib.get_option("IPDM_CONV_NO_UP2") == 1
    assert _lib.get_option("conv_no_up2") == 0
    import pytest
    with pytest.raises(_lib.IpdmError, match="unknown option"):
        _lib.set_option("no_such_switch", 1)


def test_option_values_are_range_checked():
    """ipdm_set_option refuses values outside an option's range (status code + message, the old value stays): a packed
    graph key or a kernel's static limits must never see them."""
    from ipdm_pytorch_amd import _lib
    lib = _lib.lib()
    for name, bad in (("conv_nm", 4), ("wino2_min_tiles", -1), ("unet_transpose", 2), ("direct_max_cin", 4096), ("pw_item", 3),
                      ("conv_no_wino", 7)):
        old = _lib.get_option(name)
        rc = lib.ipdm_set_option(name.encode(), bad)
        assert rc != 0 and name.encode() in lib.ipdm_last_error(), (name, rc, lib.ipdm_last_error())
        assert _lib.get_option(name) == old
    with _lib.option("unet_transpose", 1), _lib.option("conv_nm", 2):
        assert _lib.get_option("unet_transpose") == 1 and _lib.get_option("conv_nm") == 2
    # the split-bf16 modes were retired in round 5 (slower than exact f32 since the Winograd kernels): their switches are gone
    for gone in ("conv_split", "attn_split"):
        assert lib.ipdm_set_option(gone.encode(), 0) != 0 and b"unknown option" in lib.ipdm_last_error()


def test_product_library_carries_only_the_default_path():
    """The opt-in kernel (the 16-cout MFMA form of the narrow layers) lives in a second shared object, libipdm_hip_optin.so,
    which the product library loads only when the option asks for it (csrc/optin.hip): none of its device code is in
    libipdm_hip.so, the second library exports what optin.hip binds, and it reports which copy of the product it bound to."""
    import os
    from ipdm_pytorch_amd import _lib
    d = os.path.dirname(_lib.LIB_PATH)
    prod = open(_lib.LIB_PATH, "rb").read()
    optin_path = os.path.join(d, "libipdm_hip_optin.so")
    assert os.path.isfile(optin_path)
    optin = open(optin_path, "rb").read()
    for kern in (b"conv_nm_kernel",):
        assert kern not in prod and kern in optin, kern
    for kern in (b"conv_sx_kernel", b"attention_sx_kernel"):      # retired in round 5
        assert kern not in prod and kern not in optin, kern
    for kern in (b"conv_wino2_kernel", b"conv_ws_kernel", b"attention_ws_kernel", b"conv_direct"):
        assert kern in prod, kern
    h = C.CDLL(_lib.LIB_PATH) and C.CDLL(optin_path)      # (its undefined references resolve against the product library)
    for sym in ("ipdm_optin_conv_nm_eligible", "ipdm_optin_conv2d_nm_launch", "ipdm_optin_bound_to"):
        assert getattr(h, sym) is not None
    prod_h = C.CDLL(_lib.LIB_PATH)
    h.ipdm_optin_bound_to.restype = C.c_void_p
    assert h.ipdm_optin_bound_to() == C.cast(prod_h.ipdm_last_error, C.c_void_p).value      # the copy the process uses
