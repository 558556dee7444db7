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


def test_how_the_work_list_is_dealt_to_waves(rtmi):
    """The dealing rules of the compacting pair (rtmi_paths_pixels_per_wave_for: the function the kernel evaluates,
    here on the host without a device), for a device of 256 compute units at the sizes the measurements were made
    at: (pixels per wave, neighbouring pixels?)."""
    lib = rtmi.abi.load()

    def deal(w, h, n_listed, n_frames, rpp=1, meshes=0, cus=256):
        px, nb = C.c_uint32(0), C.c_int(0)
        assert lib.rtmi_paths_pixels_per_wave_for(w, h, cus, n_listed, n_frames, rpp, meshes, C.byref(px), C.byref(nb)) == 0
        return int(px.value), bool(nb.value)

    # the spheres scene at 1080p lists 2 M pixels: 64 to a wave, spread, at every launch length but one frame
    for n in (2, 8, 64, 256):
        assert deal(1920, 1080, 2070000, n) == (64, False)
    # a one-frame launch has one item per lane and nothing to refill from: waves of like pixels, whatever the scene
    assert deal(1920, 1080, 2070000, 1) == (64, True)
    assert deal(640, 360, 65000, 1, rpp=4, meshes=2) == (64, True)
    # the reference scene at 1080p lists 0.56 M: 16 neighbours from 16 frames on (5.7 rounds of wave slots), 32 over 8
    # frames (2.8 rounds: the fewest pixels that leave four paths per lane), spread below
    assert deal(1920, 1080, 560000, 256, meshes=2) == (16, True)
    assert deal(1920, 1080, 560000, 16, meshes=2) == (16, True)
    assert deal(1920, 1080, 560000, 8, meshes=2) == (32, True)
    assert deal(1920, 1080, 560000, 2, meshes=2) == (64, False)
    # without meshes the same list is spread, with the fitted count
    assert deal(1920, 1080, 560000, 256) == (16, False)
    assert deal(1920, 1080, 560000, 32) in ((32, False), (64, False))
    assert deal(1920, 1080, 560000, 2) == (64, False)
    # 3840 x 2160: 2.24 M listed
    assert deal(3840, 2160, 2240000, 512, meshes=2) == (16, True)
    assert deal(3840, 2160, 2240000, 8, meshes=2) == (32, True)
    assert deal(3840, 2160, 2240000, 2, meshes=2) == (64, True)
    # one frame of many samples is still one item per pixel: a wave needs 64 of them
    assert deal(3840, 2160, 2240000, 1, rpp=16, meshes=2) == (64, True)
    assert deal(3840, 2160, 2240000, 2, rpp=8, meshes=2) == (32, True)
    # shorter lists: 8 neighbours at 1280 x 720 (0.25 M listed), 4 at 960 x 540 (0.14 M) and at 640 x 360 (65 k: 2.6
    # rounds, nothing makes four); too few frames for that: spread
    assert deal(1280, 720, 250000, 256, meshes=2) == (8, True)
    assert deal(1280, 720, 250000, 16, meshes=2) == (16, True)     # 2.5 rounds, and 8 to a wave would leave two paths per lane
    assert deal(1280, 720, 230000, 16, meshes=2)[1] is False       # 2.3 rounds
    assert deal(960, 540, 140000, 256, meshes=2) == (4, True)
    assert deal(640, 360, 65000, 64, meshes=2) == (4, True)
    assert deal(640, 360, 65000, 64) == (16, False)
    assert deal(640, 360, 50000, 64, meshes=2) == (16, False)      # 2 rounds at 4 to a wave
    assert deal(640, 360, 65000, 2, meshes=2) == (32, False)
    # a smaller device takes neighbours sooner
    assert deal(640, 360, 65000, 64, meshes=2, cus=32) == (16, True)
    # bad arguments
    px, nb = C.c_uint32(0), C.c_int(0)
    assert lib.rtmi_paths_pixels_per_wave_for(0, 360, 256, 1, 1, 1, 0, C.byref(px), C.byref(nb)) != 0
    assert lib.rtmi_paths_pixels_per_wave_for(64, 36, 256, 64 * 36 + 1, 1, 1, 0, C.byref(px), C.byref(nb)) != 0
    assert lib.rtmi_paths_pixels_per_wave_for(64, 36, 256, 10, 1, 1, 0, None, C.byref(nb)) != 0
