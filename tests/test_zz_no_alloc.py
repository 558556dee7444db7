"""rtmi_render_frames_noalloc (the entry point that allocates nothing) against rtmi_render_frames,
through the C ABI, bit for bit.  Collected last: its two calls share the library's static work area,
the second with a smaller image, so stale contents of the first call's work list lie behind the
second call's entries."""
import ctypes as C

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _pair(rtmi, case, variant, accel=None, scene=None, fd=None, cus=0):
    """(image, display) of the plain entry point and of the no-allocation one, same inputs."""
    sc = scene if scene is not None else cases.product_scene(case)
    fd = fd if fd is not None else cases.product_frame_data(case, sc)
    w, h = case["size"]
    out = []
    for noalloc in (False, True):
        with rtmi.Renderer(w, h, variant=variant) as r:
            if accel is not None:
                r.set_acceleration(accel)
            r.assume_compute_units(cus)
            r.upload_scene(sc)
            if case.get("seed_accum"):
                r.write_accum(cases.seed_accum(case))
            (r.render_frames_noalloc if noalloc else r.render_frames)(fd, case["first"], case["n"])
            out.append((r.read_accum(), r.read_output_rgba8()))
    return out


@pytest.mark.parametrize("variant", [2, 3, 1])
def test_two_calls_in_a_row_second_one_smaller(rtmi, variant):
    big = dict(cases.CASES["c2_spheres_bench_pose"], size=(640, 360), n=16)
    small = dict(cases.CASES["c3_mesh_bench_pose"], size=(37, 19), n=9)
    for case in (big, small):
        (want, want8), (got, got8) = _pair(rtmi, case, variant)
        same = np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert same, "%dx%d: accumulation image differs in %d pixels" % (case["size"] + (int((got != want).any(axis=-1).sum()),))
        assert np.array_equal(got8, want8)
    # the small case is also a golden fixture's scene: the plain entry point is itself pinned to the oracle
    case = cases.CASES["c3_mesh_bench_pose"]
    (want, want8), (got, got8) = _pair(rtmi, case, variant)
    gold = cases.load_golden("c3_mesh_bench_pose")
    assert cases.bits_equal(got, gold["accum"]) and np.array_equal(got8, gold["rgba8"])


def test_waves_of_neighbouring_pixels(rtmi):
    """Scheduling decisions as for a device of one compute unit: the mesh scene's work list then goes to waves of
    neighbouring pixels, as a large image's does on the real device; same bits from both entry points, and the
    golden fixture's bits."""
    case = dict(cases.CASES["c3_mesh_bench_pose"], size=(192, 108), n=24)
    with rtmi.Renderer(192, 108) as r:
        r.assume_compute_units(1)
        assert r.paths_pixels_per_wave(192 * 108 // 4, 24, case["rpp"], 2)[1] is True
    (want, want8), (got, got8) = _pair(rtmi, case, 2, cus=1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(got8, want8)
    case = cases.CASES["c3_mesh_bench_pose"]
    (want, want8), (got, got8) = _pair(rtmi, case, 2, cus=1)
    gold = cases.load_golden("c3_mesh_bench_pose")
    assert cases.bits_equal(got, gold["accum"]) and np.array_equal(got8, gold["rgba8"])


def test_hierarchy_scene_and_continued_sum(rtmi):
    """The pixel-streaming kernel takes its cursor from the same static area; then a launch that continues a sum."""
    big, pose = rtmi.scenes.stress_scene(600, (10, 10), 2, seed=3, extent=3.0)
    for size, n in (((96, 54), 3), ((48, 27), 2)):
        case = dict(size=size, first=1, n=n)
        fd = rtmi.make_frame_data(size[0], size[1], big, pose=pose, rays_per_pixel=1, max_bounce=8)
        for variant in (4, 2):
            (want, want8), (got, got8) = _pair(rtmi, case, variant, accel=rtmi.abi.ACCEL_BVH, scene=big, fd=fd)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(got8, want8)
    case = cases.CASES["reference_defaults_rpp4_continue"]
    (want, want8), (got, got8) = _pair(rtmi, case, 2)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(got8, want8)
    assert cases.bits_equal(got, cases.load_golden("reference_defaults_rpp4_continue")["accum"])


def test_two_contexts_on_their_own_streams_share_the_work_area_in_turn(rtmi):
    """Two contexts on one GPU, each on its own stream, call the no-allocation entry point back to back with no host
    synchronisation in between (20 rounds, the launches of one round queued before either has finished): the
    library orders the users of the device's one work area itself, so both images must come out as from the
    plain entry point."""
    a = cases.CASES["c2_spheres_bench_pose"]
    b = cases.CASES["c3_mesh_bench_pose"]
    sa, sb = cases.product_scene(a), cases.product_scene(b)
    size_a, size_b = (320, 180), (200, 120)
    fda = rtmi.make_frame_data(size_a[0], size_a[1], sa, rtmi.BENCH_POSE, 1, 8)
    fdb = rtmi.make_frame_data(size_b[0], size_b[1], sb, rtmi.BENCH_POSE, 1, 8)
    rounds, per = 20, 8
    want = []
    for (size, sc, fd) in ((size_a, sa, fda), (size_b, sb, fdb)):
        with rtmi.Renderer(size[0], size[1], variant=2) as r:
            r.upload_scene(sc)
            for k in range(rounds):
                r.render_frames(fd, 1 + k * per, per)
            want.append(r.read_accum())
    with rtmi.Renderer(size_a[0], size_a[1], variant=2) as ra, rtmi.Renderer(size_b[0], size_b[1], variant=2) as rb:
        ra.upload_scene(sa)
        rb.upload_scene(sb)
        for k in range(rounds):
            ra.render_frames_noalloc(fda, 1 + k * per, per)
            rb.render_frames_noalloc(fdb, 1 + k * per, per)
        got = [ra.read_accum(), rb.read_accum()]
    for g, w in zip(got, want):
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))


def test_limits_and_errors(rtmi):
    a = rtmi.abi
    case = cases.CASES["c3_mesh_bench_pose"]
    sc = cases.product_scene(case)
    # one pixel more than the static work area holds
    w, h = 2048, 1025
    assert w * h > a.NOALLOC_MAX_PIXELS
    fd = rtmi.make_frame_data(w, h, sc, rtmi.BENCH_POSE, 1, 4)
    with rtmi.Renderer(w, h) as r:
        r.upload_scene(sc)
        with pytest.raises(rtmi.RtmiError) as e:
            r.render_frames_noalloc(fd, 1, 1)
        assert e.value.code == a.ERR_UNSUPPORTED and "exceed" in str(e.value)
        r.render_frames(fd, 1, 1)            # the plain entry point still serves this size
        r.synchronize()
    # exactly the maximum is accepted
    w, h = 2048, 1024
    fd = rtmi.make_frame_data(w, h, sc, rtmi.BENCH_POSE, 1, 4)
    with rtmi.Renderer(w, h, variant=2) as r:
        r.upload_scene(sc)
        r.render_frames_noalloc(fd, 1, 2)
        got = r.read_accum()
        r.render_frames(fd, 1, 2)
        assert np.array_equal(got.view(np.uint32), r.read_accum().view(np.uint32))
        r.enable_validation(True)
        with pytest.raises(rtmi.RtmiError) as e:
            r.render_frames_noalloc(fd, 1, 1)
        assert e.value.code == a.ERR_UNSUPPORTED
        r.enable_validation(False)
        r.render_frames_noalloc(fd, 1, 1)
        with pytest.raises(rtmi.RtmiError):
            r.last_render_ms()               # the call records no timing events


def test_under_stream_capture(rtmi):
    """The callers the entry point exists for: a stream in capture mode, where an allocation (or any synchronising
    call) would invalidate the capture.  The call is captured into a HIP graph on a caller-provided stream, the
    graph is launched twice (frames 1-6 overwrite, so both launches produce the same image), and the result is the
    image of the plain entry point.  Nothing runs at capture time: the image is still untouched when the capture ends."""
    hip = C.CDLL("libamdhip64.so")
    vp = C.c_void_p
    hip.hipStreamCreate.argtypes = [C.POINTER(vp)]
    hip.hipStreamBeginCapture.argtypes = [vp, C.c_int]
    hip.hipStreamEndCapture.argtypes = [vp, C.POINTER(vp)]
    hip.hipGraphInstantiate.argtypes = [C.POINTER(vp), vp, vp, vp, C.c_size_t]
    hip.hipGraphLaunch.argtypes = [vp, vp]
    hip.hipStreamSynchronize.argtypes = [vp]
    hip.hipGraphExecDestroy.argtypes = [vp]
    hip.hipGraphDestroy.argtypes = [vp]
    hip.hipStreamDestroy.argtypes = [vp]
    case = dict(cases.CASES["c2_spheres_bench_pose"], size=(320, 180))
    sc = cases.product_scene(case)
    fd = rtmi.make_frame_data(320, 180, sc, rtmi.BENCH_POSE, 1, 8)
    with rtmi.Renderer(320, 180, variant=2) as r:
        r.upload_scene(sc)
        r.render_frames(fd, 1, 6)
        want = r.read_accum()
    for variant in (2, 1):
        with rtmi.Renderer(320, 180, variant=variant) as r:
            r.upload_scene(sc)
            stream = vp()
            assert hip.hipStreamCreate(C.byref(stream)) == 0
            r.set_stream(stream.value)
            r.synchronize()
            assert hip.hipStreamBeginCapture(stream, 2) == 0          # hipStreamCaptureModeRelaxed
            r.render_frames_noalloc(fd, 1, 6)
            graph = vp()
            rc = hip.hipStreamEndCapture(stream, C.byref(graph))
            assert rc == 0 and graph.value, "capture was invalidated (hipError %d)" % rc
            assert not r.read_accum().any()                              # nothing has run yet
            exe = vp()
            assert hip.hipGraphInstantiate(C.byref(exe), graph, None, None, 0) == 0
            for _ in range(2):
                assert hip.hipGraphLaunch(exe, stream) == 0
            assert hip.hipStreamSynchronize(stream) == 0
            got = r.read_accum()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "variant %d" % variant
            hip.hipGraphExecDestroy(exe)
            hip.hipGraphDestroy(graph)
            r.set_stream(None)
            hip.hipStreamDestroy(stream)
