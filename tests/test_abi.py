"""The C-ABI library loads, exports every symbol include/rtmi.h declares, and its records have the
reference's sizes.  No compute calls: these run without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    return sorted(set(re.findall(r"RTMI_API[^;(]*?\b(rtmi_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_a_sane_number_of_symbols():
    syms = declared_symbols()
    assert len(syms) >= 30 and "rtmi_render_frames" in syms and "rtmi_load_model" in syms


def test_library_exports_every_declared_symbol(rtmi):
    lib = rtmi.abi.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    # and the Python prototype table covers exactly the header
    assert sorted(rtmi.abi.PROTOTYPES) == declared_symbols()


def test_record_sizes(rtmi, oracle):
    a = rtmi.abi
    assert [C.sizeof(t) for t in (a.FrameData, a.Material, a.Sphere, a.Mesh, a.Triangle)] == [260, 32, 48, 80, 96]
    offs = [getattr(a.FrameData, f).offset for f, _ in a.FrameData._fields_]
    assert offs == [0, 64, 128, 144, 160, 168, 172, 176, 192, 208, 224, 240, 244, 248, 252, 256]
    assert [oracle.lib().orc_sizeof(i) for i in range(5)] == [260, 32, 48, 80, 96]


def test_version_string(rtmi):
    assert b"gfx950" in rtmi.abi.load().rtmi_version()


def test_null_context_is_rejected(rtmi):
    lib = rtmi.abi.load()
    assert lib.rtmi_render(None, None) == rtmi.abi.ERR_INVALID_ARGUMENT
    assert lib.rtmi_read_accum(None, None, 0) == rtmi.abi.ERR_INVALID_ARGUMENT
    assert lib.rtmi_destroy(None) == 0
    assert lib.rtmi_create(0, 0, 0, None) == rtmi.abi.ERR_INVALID_ARGUMENT


def test_create_without_gpu_fails_loudly(rtmi):
    """On a machine with no HIP device there is no fallback: creation reports RTMI_ERR_NO_DEVICE."""
    hip = C.CDLL("libamdhip64.so")
    n = C.c_int(0)
    if hip.hipGetDeviceCount(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(rtmi.RtmiError) as e:
        rtmi.Renderer(16, 16)
    assert e.value.code == rtmi.abi.ERR_NO_DEVICE


def test_bad_extent_rejected(rtmi):
    lib = rtmi.abi.load()
    ctx = C.c_void_p()
    assert lib.rtmi_create(0, 0, 16, C.byref(ctx)) == rtmi.abi.ERR_INVALID_ARGUMENT
    assert b"extent" in lib.rtmi_last_error(None)
