"""GPU parity tests (run on an MI355X with `-m gpu`).  Everything goes through the C ABI
(include/rtmi.h via ctypes).  The bar is bit-exactness against the CPU oracle for every output:
accumulation image (float32 bit patterns), display image, primitive ids, per-pixel id-chain
hash, RNG state and event counters."""
import ctypes as C

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

VARIANTS = [0, 1, 2, 3, 4]   # 3 = chosen per launch


def gpu_render(rtmi, case, variant, validate=True, per_frame=False):
    sc = cases.product_scene(case)
    fd = cases.product_frame_data(case, sc)
    w, h = case["size"]
    with rtmi.Renderer(w, h, variant=variant) as r:
        r.upload_scene(sc)
        r.enable_validation(validate)
        if case.get("seed_accum"):
            r.write_accum(cases.seed_accum(case))
        if per_frame:
            for f in range(case["first"], case["first"] + case["n"]):
                fd.frame_index = f
                r.render(fd)
        else:
            r.render_frames(fd, case["first"], case["n"])
        out = dict(accum=r.read_accum(), rgba8=r.read_output_rgba8())
        if validate:
            out.update(prim_id=r.read_hit_ids(), chain_hash=r.read_chain_hash(), rng_state=r.read_rng_state(),
                       counters=r.counters())
    return out


def assert_same(got, want, keys=("rgba8", "prim_id", "chain_hash", "rng_state")):
    # booleans are computed first so a failure prints a short message, not the arrays
    same = cases.bits_equal(got["accum"], want["accum"])
    assert same, "accumulation image differs: %d pixels, max abs %g" % (
        int((got["accum"] != want["accum"]).any(axis=-1).sum()), float(np.nanmax(np.abs(got["accum"] - want["accum"]))))
    for k in keys:
        if k in got:
            same = np.array_equal(got[k], want[k])
            assert same, "%s differs in %d pixels" % (k, int((got[k] != want[k]).sum()))
    if "counters" in got:
        same = got["counters"] == want["counters"]
        assert same, "counters differ: %r vs %r" % (got["counters"], want["counters"])


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_golden_and_oracle(rtmi, oracle, name, variant):
    case = cases.CASES[name]
    got = gpu_render(rtmi, case, variant)
    assert_same(got, cases.load_golden(name))
    assert_same(got, cases.run_oracle(case))


@pytest.mark.parametrize("variant", VARIANTS)
def test_host_sides_agree(rtmi, oracle, variant):
    """The product's scene/camera bytes equal the oracle's, so both render the same inputs."""
    for case in cases.CASES.values():
        sc, osc = cases.product_scene(case), cases.oracle_scene(case)
        assert bytes(cases.product_frame_data(case, sc)) == bytes(cases.oracle_frame_data(case, osc))


@pytest.mark.parametrize("variant", VARIANTS)
def test_frame_loop_equals_batched_launch(rtmi, variant):
    """rtmi_render called once per frame (the reference's loop) == one rtmi_render_frames launch."""
    for name in ("c3_mesh_bench_pose", "reference_defaults_rpp4_continue"):
        case = cases.CASES[name]
        a = gpu_render(rtmi, case, variant, per_frame=True)
        b = gpu_render(rtmi, case, variant, per_frame=False)
        assert cases.bits_equal(a["accum"], b["accum"]) and np.array_equal(a["rgba8"], b["rgba8"])
        assert np.array_equal(a["rng_state"], b["rng_state"]) and np.array_equal(a["prim_id"], b["prim_id"])
        assert_same(b, cases.load_golden(name), keys=("rgba8", "prim_id", "rng_state"))


@pytest.mark.parametrize("variant", VARIANTS)
def test_uninstrumented_kernels_give_the_same_image(rtmi, variant):
    for name in ("c2_spheres_bench_pose", "ragged_size_rpp3"):
        got = gpu_render(rtmi, cases.CASES[name], variant, validate=False)
        want = cases.load_golden(name)
        assert cases.bits_equal(got["accum"], want["accum"]) and np.array_equal(got["rgba8"], want["rgba8"])


def test_reset_and_accumulate_semantics(rtmi):
    case = cases.CASES["c3_mesh_bench_pose"]
    sc = cases.product_scene(case)
    fd = cases.product_frame_data(case, sc)
    w, h = case["size"]
    with rtmi.Renderer(w, h) as r:
        r.upload_scene(sc)
        r.write_accum(np.full((h, w, 4), 55.0, dtype=np.float32))
        r.render_frames(fd, 1, 3)                 # frame_index 1 overwrites the junk
        a = r.read_accum()
        r.render_frames(fd, 4, 5)                 # later frames add
        b = r.read_accum()
        r.render_frames(fd, 1, 8)                 # "camera moved": starts over
        c = r.read_accum()
        r.clear_accum()
        assert not r.read_accum().any() and not r.read_output_rgba8().any()
    want = cases.load_golden("c3_mesh_bench_pose")
    assert cases.bits_equal(b, want["accum"]) and cases.bits_equal(c, want["accum"])
    assert not cases.bits_equal(a, b) and (a[..., 3] == 1.0).all()


def test_display_image_is_sum_over_frame_index(rtmi):
    case = cases.CASES["c2_spheres_bench_pose"]
    got = gpu_render(rtmi, case, 1, validate=False)
    mean = got["accum"][..., :3] / np.float32(case["first"] + case["n"] - 1)
    want = np.rint(np.clip(mean, 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
    assert np.array_equal(got["rgba8"][..., :3], want) and (got["rgba8"][..., 3] == 255).all()


def test_resolve_output(rtmi):
    case = cases.CASES["c3_mesh_bench_pose"]
    sc = cases.product_scene(case)
    fd = cases.product_frame_data(case, sc)
    w, h = case["size"]
    with rtmi.Renderer(w, h) as r:
        r.upload_scene(sc)
        r.render_frames(fd, 1, 8)
        acc = r.read_accum()
        acc[..., 3] = 2.0                          # as if two partial images had been summed
        r.write_accum(acc * np.float32(2.0))
        r.resolve_output(16)
        out, fixed = r.read_output_rgba8(), r.read_accum()
    assert np.array_equal(out, cases.load_golden("c3_mesh_bench_pose")["rgba8"])
    assert (fixed[..., 3] == 1.0).all() and np.array_equal(fixed[..., :3], acc[..., :3] * np.float32(2.0))


@pytest.mark.parametrize("variant", VARIANTS)
def test_edge_shapes_and_empty_scenes(rtmi, oracle, variant):
    base = cases.CASES["c3_mesh_bench_pose"]
    for size in ((1, 1), (17, 9), (16, 16), (15, 33)):
        case = dict(base, size=size, n=2)
        assert_same(gpu_render(rtmi, case, variant), cases.run_oracle(case))
    # no spheres and no meshes in the frame data: every path is a sky path
    case = dict(base, size=(24, 12), n=2)
    sc, osc = cases.product_scene(case), cases.oracle_scene(case)
    fd, ofd = cases.product_frame_data(case, sc), cases.oracle_frame_data(case, osc)
    for ns, nm in ((0, 0), (1, 0), (0, 2), (0, 1)):
        fd.sphere_number, fd.mesh_number = ns, nm
        ofd.sphere_count, ofd.mesh_count = ns, nm
        with rtmi.Renderer(24, 12, variant=variant) as r:
            r.upload_scene(sc)
            r.enable_validation(True)
            r.render_frames(fd, 1, 2)
            got = dict(accum=r.read_accum(), rgba8=r.read_output_rgba8(), prim_id=r.read_hit_ids(),
                       chain_hash=r.read_chain_hash(), rng_state=r.read_rng_state(), counters=r.counters())
        assert_same(got, oracle.render(ofd, osc, 1, 2))
    # an uploaded scene with no primitives at all
    empty = rtmi.Scene()
    fd.sphere_number = fd.mesh_number = 0
    ofd.sphere_count = ofd.mesh_count = 0
    with rtmi.Renderer(24, 12, variant=variant) as r:
        r.upload_scene(empty)
        r.render_frames(fd, 1, 2)
        assert cases.bits_equal(r.read_accum(), oracle.render(ofd, osc, 1, 2)["accum"])


@pytest.mark.parametrize("variant", [0, 2])
def test_duplicated_triangles_and_overlapping_meshes(rtmi, oracle, variant):
    """Every triangle of the reference scene twice in a row (each hit is then a tie between two
    neighbouring indices, which the wave-shared triangle tests hand to different lanes) and the cube's
    mesh record listed a second time: the lowest index of the first mesh must win every time."""
    case = dict(cases.CASES["c3_mesh_bench_pose"], size=(80, 45), n=3)
    base = cases.product_scene(case)
    ns, nt, nm = base.n_spheres, base.n_triangles, base.n_meshes
    sph = np.frombuffer(base.sphere_array(), dtype=rtmi.host.SPHERE_DTYPE, count=ns)
    tri = np.frombuffer(base.triangles, dtype=rtmi.host.TRIANGLE_DTYPE, count=nt)
    msh = np.frombuffer(base.meshes, dtype=rtmi.host.MESH_DTYPE, count=nm)
    tri2 = np.repeat(tri, 2)                                   # t0 t0 t1 t1 ...
    msh2 = np.concatenate([msh, msh[-1:]]).copy()
    msh2["start_triangle"] = np.concatenate([2 * msh["start_triangle"], 2 * msh["start_triangle"][-1:]])
    msh2["num_triangles"] = np.concatenate([2 * msh["num_triangles"], 2 * msh["num_triangles"][-1:]])
    sc = rtmi.Scene.from_records(sph, tri2, msh2)
    osc = oracle.Scene.from_product(sc)
    w, h = case["size"]
    fd = rtmi.make_frame_data(w, h, sc, pose=rtmi.BENCH_POSE, rays_per_pixel=2, max_bounce=8)
    pos, yaw, pitch = rtmi.BENCH_POSE
    ofd = oracle.frame_data(w, h, pos=pos, yaw=yaw, pitch=pitch, rpp=2, max_bounce=8, scene=osc)
    want = oracle.render(ofd, osc, 1, 3)
    with rtmi.Renderer(w, h, variant=variant) as r:
        r.upload_scene(sc)
        assert r.acceleration() == rtmi.abi.ACCEL_NONE
        r.enable_validation(True)
        r.render_frames(fd, 1, 3)
        got = dict(accum=r.read_accum(), rgba8=r.read_output_rgba8(), prim_id=r.read_hit_ids(),
                   chain_hash=r.read_chain_hash(), rng_state=r.read_rng_state(), counters=r.counters())
    assert_same(got, want)
    tri_hits = want["prim_id"][(want["prim_id"] != oracle.MISS_ID) & (want["prim_id"] >= ns)] - ns
    assert len(tri_hits) > 300 and (tri_hits % 2 == 0).all()       # always the first copy
    assert (tri_hits < 2 * nt).all()


@pytest.mark.parametrize("variant", VARIANTS)
def test_zero_bounces_and_many_samples(rtmi, oracle, variant):
    base = cases.CASES["c3_mesh_bench_pose"]
    for mb, rpp in ((0, 1), (1, 1), (2, 7), (16, 2)):
        case = dict(base, size=(40, 24), max_bounce=mb, rpp=rpp, n=2)
        assert_same(gpu_render(rtmi, case, variant), cases.run_oracle(case))


def test_large_frame_indices(rtmi, oracle):
    # frame indices as config C4 uses them (last eighth of 4096 frames), seed arithmetic wraps in uint32
    case = dict(cases.CASES["c3_mesh_bench_pose"], size=(48, 27), first=3585, n=3)
    assert_same(gpu_render(rtmi, case, 1), cases.run_oracle(case))


def test_error_paths(rtmi):
    a = rtmi.abi
    case = cases.CASES["c3_mesh_bench_pose"]
    sc = cases.product_scene(case)
    fd = cases.product_frame_data(case, sc)
    w, h = case["size"]
    with rtmi.Renderer(w, h) as r:
        with pytest.raises(rtmi.RtmiError) as e:
            r.render_frames(fd, 1, 1)
        assert e.value.code == a.ERR_NO_SCENE
        r.upload_scene(sc)
        fd.sphere_number = 5
        with pytest.raises(rtmi.RtmiError) as e:
            r.render_frames(fd, 1, 1)
        assert e.value.code == a.ERR_SCENE_MISMATCH
        fd.sphere_number = 1
        fd.window[0] = w + 1
        with pytest.raises(rtmi.RtmiError) as e:
            r.render_frames(fd, 1, 1)
        assert e.value.code == a.ERR_INVALID_ARGUMENT
        fd.window[0] = w
        with pytest.raises(rtmi.RtmiError) as e:
            r.set_variant(7)
        assert e.value.code == a.ERR_UNSUPPORTED
        small = np.zeros(4, dtype=np.float32)
        assert a.load().rtmi_read_accum(r._ctx, small.ctypes.data, small.nbytes) == a.ERR_INVALID_ARGUMENT
        with pytest.raises(rtmi.RtmiError):
            r.read_hit_ids()                      # validation outputs were never enabled
        bad = rtmi.Scene()
        bad._meshes[0].start_triangle, bad._meshes[0].num_triangles = 0, 3
        bad._n_meshes.value = 1
        with pytest.raises(rtmi.RtmiError) as e:
            r.upload_scene(bad)                   # mesh names triangles that do not exist
        assert e.value.code == a.ERR_INVALID_ARGUMENT
        for first, n in ((1, (1 << 24) + 1), (0xFFFFFFFF, 2)):
            with pytest.raises(rtmi.RtmiError) as e:
                r.render_frames(fd, first, n)     # more frames than a call takes; frame index past 32 bits
            assert e.value.code == a.ERR_INVALID_ARGUMENT
        r.render_frames(fd, 0xFFFFFFFF, 1)        # the last frame index there is
        r.render_frames(fd, 1, 1)                 # the context is still usable
        r.synchronize()
    ctx = C.c_void_p()
    assert a.load().rtmi_create(99, 8, 8, C.byref(ctx)) == a.ERR_INVALID_ARGUMENT


def test_device_arithmetic_contract(rtmi, oracle):
    """The primitives the contract relies on: IEEE divide / sqrt / reciprocal, exact uint->float
    scaling, and the polynomial log2 / exp2 / pow, against numpy and the oracle bit for bit."""
    rs = np.random.RandomState(99)
    bits = rs.randint(0, 2**32, size=1 << 18, dtype=np.uint64).astype(np.uint32)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, 1e-39, 1.17549435e-38, 3.4028235e38,
                        0.5, 2.0, 3.0, 1e-6, 0.0001, 1.0000001, 0.99999994], dtype=np.float32)
    x = np.concatenate([bits.view(np.float32), special, rs.uniform(-4, 4, 4096).astype(np.float32)])
    y = np.concatenate([np.roll(bits, 7).view(np.float32), special[::-1], rs.uniform(-4, 4, 4096).astype(np.float32)])
    L = oracle.lib()
    with rtmi.Renderer(8, 8) as r, np.errstate(all="ignore"):
        assert cases.bits_equal(r.selftest_math(0, x), np.float32(1.0) / x)
        assert cases.bits_equal(r.selftest_math(1, x), np.sqrt(x))
        assert cases.bits_equal(r.selftest_math(6, x, y), x / y)
        u = x.view(np.uint32)
        assert cases.bits_equal(r.selftest_math(4, x), u.astype(np.float32) * np.float32(2.0 ** -32))
        n = 6000
        xs = np.concatenate([x[:n], special, rs.uniform(0, 1, n).astype(np.float32), rs.uniform(-160, 140, n).astype(np.float32)])
        ys = np.concatenate([y[:n], special[::-1], np.full(n, 0.35, dtype=np.float32), rs.uniform(0, 8, n).astype(np.float32)])
        want_log = np.array([L.orc_log2(float(v)) for v in xs], dtype=np.float32)
        want_exp = np.array([L.orc_exp2(float(v)) for v in xs], dtype=np.float32)
        want_pow = np.array([L.orc_pow(float(a), float(b)) for a, b in zip(xs, ys)], dtype=np.float32)
        assert cases.bits_equal(r.selftest_math(2, xs), want_log)
        assert cases.bits_equal(r.selftest_math(3, xs), want_exp)
        assert cases.bits_equal(r.selftest_math(5, xs, ys), want_pow)


def test_external_buffer_and_stream(rtmi):
    """Accumulate into a torch tensor on torch's stream (the multi-GPU path's plumbing)."""
    import torch
    case = cases.CASES["c3_mesh_bench_pose"]
    sc = cases.product_scene(case)
    fd = cases.product_frame_data(case, sc)
    w, h = case["size"]
    dev = torch.device("cuda:0")
    from raytracer_amd.distributed import ShardedRenderer
    with rtmi.Renderer(w, h) as r:
        r.upload_scene(sc)
        sr = ShardedRenderer(r, dev)
        first, n = sr.render(fd, 8)
        assert (first, n) == (1, 8)
        sr.resolve(8)
        got = sr.accum.cpu().numpy()      # on torch's current stream, which render() / resolve() joined: no device-wide wait
        out = r.read_output_rgba8()
        sr.close()
    want = cases.load_golden("c3_mesh_bench_pose")
    assert cases.bits_equal(got, want["accum"]) and np.array_equal(out, want["rgba8"])


# ---- Benchmark sizes: 1920x1080, depth 8 -----------------------------------------------------------

FULL = dict(size=(1920, 1080), rpp=1, max_bounce=8, first=1)


@pytest.mark.parametrize("scene", ["spheres", "reference"])
def test_full_size_rows_against_oracle(rtmi, oracle, scene):
    """At the benchmark resolution the oracle checks whole rows (a pixel depends on no other pixel,
    so rows of a full render can be checked on their own): bottom, horizon, middle and top rows."""
    case = dict(FULL, scene=scene, pose=cases.BENCH_POSE, n=6)
    got = gpu_render(rtmi, case, 2)
    for other in (0, 1):
        again = gpu_render(rtmi, case, other)
        assert cases.bits_equal(got["accum"], again["accum"]) and got["counters"] == again["counters"]
        assert np.array_equal(got["chain_hash"], again["chain_hash"]) and np.array_equal(got["rng_state"], again["rng_state"])
    osc = cases.oracle_scene(case)
    ofd = cases.oracle_frame_data(case, osc)
    for y0 in (0, 300, 539, 1079):
        want = oracle.render(ofd, osc, 1, 6, rows=(y0, y0 + 1))
        for k in ("accum", "rgba8", "prim_id", "chain_hash", "rng_state"):
            assert np.array_equal(got[k][y0].view(np.uint32) if k == "accum" else got[k][y0],
                                  want[k][y0].view(np.uint32) if k == "accum" else want[k][y0]), (k, y0)
    c = got["counters"]
    npx = 1920 * 1080
    assert c["n_pixel_frames"] == npx * 6 and c["n_paths"] == npx * 6
    assert c["n_segments"] == c["n_hit_bounces"] + c["n_env_evals"]
    assert c["n_hit_bounces"] == c["n_sphere_wins"] + c["n_tri_wins"]
    assert c["n_sphere_tests"] == c["n_segments"] * len(cases.sphere_table() if scene == "spheres" else [0])
    assert c["n_env_evals"] <= c["n_paths"]


def test_full_size_split_invariance_256_frames(rtmi):
    """Benchmark configuration (1080p, 256 frames, depth 8): one launch == four launches of 64 frames ==
    the other variant, bit for bit; a second run reproduces the first."""
    case = dict(FULL, scene="spheres", pose=cases.BENCH_POSE)
    sc = cases.product_scene(case)
    fd = cases.product_frame_data(case, sc)
    with rtmi.Renderer(1920, 1080, variant=2) as r:
        r.upload_scene(sc)
        r.render_frames(fd, 1, 256)
        one = r.read_accum()
        out_one = r.read_output_rgba8()
        for k in range(4):
            r.render_frames(fd, 1 + 64 * k, 64)
        four = r.read_accum()
        out_four = r.read_output_rgba8()
        assert np.array_equal(one.view(np.uint32), four.view(np.uint32)) and np.array_equal(out_one, out_four)
        for other in (0, 1):
            r.set_variant(other)
            r.render_frames(fd, 1, 256)
            same = np.array_equal(one.view(np.uint32), r.read_accum().view(np.uint32))
            assert same, "variant %d differs from variant 2" % other
    assert np.isfinite(one).all() and (one[..., 3] == 1.0).all()
    mean = one[..., :3] / 256.0
    assert 0.2 < mean.mean() < 2.0


def _deal(rtmi, size, n_listed, n_frames, mesh_number=0):
    """(pixels per wave, neighbours in the list?) of the compacting pair for such a launch"""
    with rtmi.Renderer(size[0], size[1]) as r:
        return r.paths_pixels_per_wave(n_listed, n_frames, 1, mesh_number)


def test_work_list_dealt_32_pixels_to_a_wave(rtmi, oracle):
    """The compacting pair deals its work list 64, 32 or 16 pixels to a wave, by the list's length and the
    launch's frame count (rtmi_paths_pixels_per_wave tells which): the benchmark image takes 64, the reference scene
    at 1080p over 256 frames and the small test images over enough frames 16.  This image over 200 frames lands in
    between, with more frames than the 16 its ring then holds: bit for bit against the one-lane-per-pixel kernel, and
    rows of it against the oracle."""
    case = dict(scene="spheres", pose=cases.BENCH_POSE, size=(1600, 900), rpp=1, max_bounce=8, first=1, n=200)
    got = gpu_render(rtmi, case, 2)
    n_hit = int((got["prim_id"] != rtmi.abi.MISS_ID).sum())
    assert _deal(rtmi, case["size"], n_hit, case["n"]) == (32, False), n_hit
    again = gpu_render(rtmi, case, 1)
    assert cases.bits_equal(got["accum"], again["accum"]) and got["counters"] == again["counters"]
    assert np.array_equal(got["chain_hash"], again["chain_hash"]) and np.array_equal(got["rng_state"], again["rng_state"])
    osc = cases.oracle_scene(case)
    ofd = cases.oracle_frame_data(case, osc)
    for y0 in (0, 450, 899):
        want = oracle.render(ofd, osc, 1, case["n"], rows=(y0, y0 + 1))
        assert np.array_equal(got["accum"][y0].view(np.uint32), want["accum"][y0].view(np.uint32)), y0
        assert np.array_equal(got["chain_hash"][y0], want["chain_hash"][y0]), y0


def test_pixels_per_wave_by_list_length_and_frame_count(rtmi, oracle):
    """The same image and scene rendered over 1, 6 and 96 frames is dealt to waves in different ways (few frames would
    leave lanes of a short wave without items; many frames make more, shorter waves worth their set-up): each against
    the oracle, whole image, every output.  And what the choice does at the benchmark sizes."""
    seen = set()
    for n in (1, 6, 96):
        case = dict(scene="reference", pose=cases.BENCH_POSE, size=(640, 360), rpp=1, max_bounce=4, first=1, n=n)
        got = gpu_render(rtmi, case, 2)
        n_hit = int((got["prim_id"] != rtmi.abi.MISS_ID).sum())
        seen.add(_deal(rtmi, case["size"], n_hit, n, 2))
        osc = cases.oracle_scene(case)
        want = oracle.render(cases.oracle_frame_data(case, osc), osc, 1, n)
        assert_same({k: v for k, v in got.items() if k != "counters"}, want)
    assert len(seen) >= 2 and (64, True) in seen, seen     # one frame: one item per lane, waves of like pixels
    # 1080p: the spheres scene lists every pixel below the horizon and more (5.3 rounds of wave slots at 64 to a wave)
    assert _deal(rtmi, (1920, 1080), 2000000, 256) == (64, False)
    # the reference scene lists about 0.56 M pixels: 16 neighbours to a wave over 256 frames, 32 over 8, 64 spread ones over 2
    assert _deal(rtmi, (1920, 1080), 560000, 256, 2) == (16, True)
    assert _deal(rtmi, (1920, 1080), 560000, 8, 2) == (32, True)
    assert _deal(rtmi, (1920, 1080), 560000, 2, 2) == (64, False)
    assert _deal(rtmi, (1920, 1080), 560000, 256, 0) == (16, False)   # without meshes: spread (neighbours only in one-frame launches)


@pytest.mark.parametrize("size,n,want_deal,rows", [((3840, 2160), 8, (32, True), (0, 1100, 2159)),
                                                   ((1280, 720), 64, (8, True), (0, 400, 719))])
def test_waves_of_neighbouring_pixels(rtmi, oracle, size, n, want_deal, rows):
    """With meshes and a list that fills the device's wave slots four times or more, a wave takes neighbouring list
    entries: the reference scene at 3840x2160 over 8 frames 32 to a wave, at 1280x720 over 64 frames 8 to a wave (16
    would make too few waves).  Bit for bit against the one-lane-per-pixel kernel, rows against the oracle.
    (16 neighbours: the 256-frame test of the reference scene at 1080p; 64: the four-frame test at 3840x2160.)"""
    case = dict(scene="reference", pose=cases.BENCH_POSE, size=size, rpp=1, max_bounce=8, first=1, n=n)
    got = gpu_render(rtmi, case, 2)
    n_hit = int((got["prim_id"] != rtmi.abi.MISS_ID).sum())
    assert _deal(rtmi, size, n_hit, n, 2) == want_deal, n_hit
    if size == (3840, 2160):
        assert _deal(rtmi, size, n_hit, 4, 2) == (64, True)
        assert _deal(rtmi, (1920, 1080), n_hit // 4, 256, 2) == (16, True)
    again = gpu_render(rtmi, case, 1)
    assert cases.bits_equal(got["accum"], again["accum"]) and got["counters"] == again["counters"]
    assert np.array_equal(got["chain_hash"], again["chain_hash"]) and np.array_equal(got["rng_state"], again["rng_state"])
    del again
    osc = cases.oracle_scene(case)
    ofd = cases.oracle_frame_data(case, osc)
    for y0 in rows:
        want = oracle.render(ofd, osc, 1, n, rows=(y0, y0 + 1))
        for k in ("accum", "rgba8", "prim_id", "chain_hash", "rng_state"):
            same = np.array_equal(got[k][y0].view(np.uint32) if k == "accum" else got[k][y0],
                                  want[k][y0].view(np.uint32) if k == "accum" else want[k][y0])
            assert same, (k, y0)


@pytest.mark.parametrize("size,n,cus,want_deal", [((128, 72), 24, 1, (16, True)), ((128, 72), 32, 2, (8, True)),
                                                  ((256, 144), 4, 1, (64, True)), ((256, 144), 8, 2, (32, True)), ((128, 72), 64, 4, (4, True)),
                                                  ((128, 72), 24, 64, (16, False)), ((128, 72), 24, 0, None)])
def test_every_way_of_dealing_work_on_a_device_taken_to_be_small(rtmi, oracle, size, n, cus, want_deal):
    """rtmi_assume_compute_units makes the scheduling decisions for a device of that many compute units, so that a
    small image goes through the forms that only large ones take on the real device: waves of 64, 32, 16, 8 and 4
    neighbouring pixels, and the spread form.  Whole image, every output, against the oracle; the event counters
    against the one-lane-per-pixel kernel."""
    case = dict(scene="reference", pose=cases.BENCH_POSE, size=size, rpp=1, max_bounce=6, first=1, n=n)
    sc = cases.product_scene(case)
    fd = cases.product_frame_data(case, sc)
    osc = cases.oracle_scene(case)
    want = oracle.render(cases.oracle_frame_data(case, osc), osc, 1, n)
    n_hit = int((want["prim_id"] != rtmi.abi.MISS_ID).sum())
    got = {}
    for variant in (2, 1):
        with rtmi.Renderer(size[0], size[1], variant=variant) as r:
            r.assume_compute_units(cus)
            if variant == 2 and want_deal is not None:
                assert r.paths_pixels_per_wave(n_hit, n, 1, 2) == want_deal, n_hit
            r.upload_scene(sc)
            r.enable_validation(True)
            r.render_frames(fd, 1, n)
            got[variant] = dict(accum=r.read_accum(), rgba8=r.read_output_rgba8(), prim_id=r.read_hit_ids(),
                                chain_hash=r.read_chain_hash(), rng_state=r.read_rng_state(), counters=r.counters())
    assert got[2]["counters"] == got[1]["counters"]
    assert_same({k: v for k, v in got[2].items() if k != "counters"}, want)


def test_sphere_candidates_evaluated_in_batches(rtmi, oracle):
    """The sphere-only form of the compacting pair stashes a lane's candidate sphere and evaluates the stashed ones
    of a wave together (when a lane holding one finds another, and at the end).  Three scenes that drive that
    differently: 60 nested shells around the camera's target (every ray that reaches them is a candidate of each:
    a flush per sphere), 150 small spheres far apart (hardly ever two for one ray: the final flush does the work),
    and pairs of equal spheres at the same place (ties between candidates, which must fall to the lower index).
    Whole images, every output, bit for bit against the oracle and against the one-lane-per-pixel kernel."""
    rs = np.random.RandomState(77)
    grey = lambda k: rtmi.material((0.5 + 0.4 * np.sin(k), 0.6, 0.5 + 0.4 * np.cos(k)), 2.0 if k % 9 == 0 else 0.0, 0.3 * (k % 3))
    shells = [rtmi.sphere((0.0, 0.0, 0.0), 0.2 + 0.05 * k, grey(k)) for k in range(60)]
    sparse = [rtmi.sphere(tuple(float(v) for v in rs.uniform(-12, 12, 3)), 0.25, grey(k)) for k in range(150)]
    twins = []
    for k in range(20):
        c = tuple(float(v) for v in rs.uniform(-3, 3, 3))
        twins += [rtmi.sphere(c, 0.7, grey(2 * k)), rtmi.sphere(c, 0.7, grey(2 * k + 1))]
    for name, spheres, pos in (("shells", shells, (0.0, 0.5, 6.0)), ("sparse", sparse, (0.0, 0.0, 20.0)), ("twins", twins, (0.0, 1.0, 8.0))):
        sc = rtmi.Scene(spheres)
        osc = oracle.Scene.from_product(sc)
        w, h = 160, 96
        fd = rtmi.make_frame_data(w, h, sc, pose=(pos, -90.0, 0.0), rays_per_pixel=2, max_bounce=5)
        ofd = oracle.frame_data(w, h, pos=pos, yaw=-90.0, pitch=0.0, rpp=2, max_bounce=5, scene=osc)
        want = oracle.render(ofd, osc, 1, 5)
        assert (want["prim_id"] != rtmi.abi.MISS_ID).sum() > 200, name
        for variant in (2, 0):
            with rtmi.Renderer(w, h, variant=variant) as r:
                r.upload_scene(sc)
                r.enable_validation(True)
                r.render_frames(fd, 1, 5)
                got = dict(accum=r.read_accum(), rgba8=r.read_output_rgba8(), prim_id=r.read_hit_ids(),
                           chain_hash=r.read_chain_hash(), rng_state=r.read_rng_state())
            assert_same(got, want)


def test_full_size_mesh_scene_256_frames(rtmi):
    """Config C3 at its stated length (reference scene, 1080p, 256 frames, depth 8): one launch == four
    launches of 64 frames == the one-lane-per-pixel kernel, bit for bit.  256 frames take the wave-shared
    triangle tests through the 8-frame ring 32 times over."""
    case = dict(FULL, scene="reference", pose=cases.BENCH_POSE)
    sc = cases.product_scene(case)
    fd = cases.product_frame_data(case, sc)
    with rtmi.Renderer(1920, 1080, variant=2) as r:
        r.upload_scene(sc)
        r.render_frames(fd, 1, 256)
        one = r.read_accum()
        out_one = r.read_output_rgba8()
        for k in range(4):
            r.render_frames(fd, 1 + 64 * k, 64)
        four = r.read_accum()
        assert np.array_equal(one.view(np.uint32), four.view(np.uint32)) and np.array_equal(out_one, r.read_output_rgba8())
        r.set_variant(1)
        r.render_frames(fd, 1, 256)
        same = np.array_equal(one.view(np.uint32), r.read_accum().view(np.uint32))
        assert same, "variant 1 differs from variant 2"
        r.set_variant(3)                  # chosen per launch, primary segments kept between the two launches
        r.render_frames(fd, 1, 128)
        r.render_frames(fd, 129, 128)
        same = np.array_equal(one.view(np.uint32), r.read_accum().view(np.uint32))
        assert same, "two launches under variant 3 differ from one launch of variant 2"
    assert np.isfinite(one).all() and (one[..., 3] == 1.0).all()
    assert 0.2 < (one[..., :3] / 256.0).mean() < 2.0


def test_4k_last_rank_frame_range(rtmi, oracle):
    """Config C4's image (reference scene, 3840x2160, depth 8) on the frame range its eighth rank starts
    with (frames 3585..3588 of 4096): the float part of the seed (Raytracing.comp:462) reaches
    3839 + 3840 * 2159 = 8 294 399 (exact in binary32, close to 2^23) and the frame term wraps in uint32.
    Variant 2 against variants 0 and 1 bit for bit, four whole rows against the oracle, counter identities."""
    w, h, first, n = 3840, 2160, 3585, 4
    case = dict(scene="reference", pose=cases.BENCH_POSE, size=(w, h), rpp=1, max_bounce=8, first=first, n=n)
    sc = cases.product_scene(case)
    fd = cases.product_frame_data(case, sc)
    got = {}
    with rtmi.Renderer(w, h, variant=2) as r:
        r.upload_scene(sc)
        r.enable_validation(True)
        for variant in (2, 0, 1):
            r.set_variant(variant)
            r.clear_accum()
            r.reset_counters()
            r.render_frames(fd, first, n)
            cur = dict(accum=r.read_accum(), rgba8=r.read_output_rgba8(), prim_id=r.read_hit_ids(),
                       chain_hash=r.read_chain_hash(), rng_state=r.read_rng_state(), counters=r.counters())
            if variant == 2:
                got = cur
                continue
            for k in ("accum", "rgba8", "prim_id", "chain_hash", "rng_state"):
                same = np.array_equal(got[k].view(np.uint32) if k == "accum" else got[k],
                                      cur[k].view(np.uint32) if k == "accum" else cur[k])
                assert same, "variant %d differs from variant 2 in %s" % (variant, k)
            assert cur["counters"] == got["counters"]
            del cur
    osc = cases.oracle_scene(case)
    ofd = cases.oracle_frame_data(case, osc)
    for y0 in (0, 700, 1079, 2159):
        want = oracle.render(ofd, osc, first, n, rows=(y0, y0 + 1))
        for k in ("accum", "rgba8", "prim_id", "chain_hash", "rng_state"):
            same = np.array_equal(got[k][y0].view(np.uint32) if k == "accum" else got[k][y0],
                                  want[k][y0].view(np.uint32) if k == "accum" else want[k][y0])
            assert same, (k, y0)
        del want
    # the seed of the last pixel, frame 3588: uint(3839 + 3840 * 2159) + 3588 * 719393 (mod 2^32), advanced by the
    # path's random numbers; its frame-start value is what the oracle row above already pins, here the formula itself
    assert float(np.float32(3839.0) + np.float32(3840.0) * np.float32(2159.0)) == 8294399.0
    c = got["counters"]
    npx = w * h
    assert c["n_pixel_frames"] == npx * n and c["n_paths"] == npx * n
    assert c["n_segments"] == c["n_hit_bounces"] + c["n_env_evals"]
    assert c["n_hit_bounces"] == c["n_sphere_wins"] + c["n_tri_wins"]
    assert c["n_sphere_tests"] == c["n_segments"] and c["n_box_tests"] == 2 * c["n_segments"]
    assert c["n_env_evals"] <= c["n_paths"]


# ---- ClosestHit on hand-made rays -----------------------------------------------------------------

def _oracle_sphere_scene(oracle, table):
    return oracle.Scene([oracle.make_sphere(c, r, oracle.material((1, 1, 1), 0.0, 0.0)) for c, r in table])


def _product_sphere_scene(rtmi, table):
    return rtmi.Scene([rtmi.sphere(c, r, rtmi.material((1, 1, 1))) for c, r in table])


def _check_rays(rtmi, oracle, sc, osc, o, d, ns, nm):
    with rtmi.Renderer(8, 8) as r:
        r.upload_scene(sc)
        t, prim = r.selftest_closest_hit(ns, nm, o, d)
    wt, wprim = oracle.closest_hit(osc, o, d, ns, nm)
    same_t = cases.bits_equal(t, wt)
    same_p = np.array_equal(prim, wprim)
    assert same_p, "prim ids differ for %d rays" % int((prim != wprim).sum())
    assert same_t, "t differs for %d rays" % int((t.view(np.uint32) != wt.view(np.uint32)).sum())
    return t, prim


def test_closest_hit_analytic(rtmi, oracle):
    table = [((0.0, 0.0, 0.0), 1.0)]
    sc, osc = _product_sphere_scene(rtmi, table), _oracle_sphere_scene(oracle, table)
    o = np.array([[0, 0, -3], [0, 0, -3], [0, 0, 0], [0, 0, -3], [0, 1.5, -3]], dtype=np.float32)
    d = np.array([[0, 0, 1], [0, 0, 0.5], [0, 0, 1], [0, 0, -1], [0, 0, 1]], dtype=np.float32)
    t, prim = _check_rays(rtmi, oracle, sc, osc, o, d, 1, 0)
    assert list(t[:2]) == [2.0, 4.0] and list(prim) == [0, 0, rtmi.abi.MISS_ID, rtmi.abi.MISS_ID, rtmi.abi.MISS_ID]
    assert t[2] == np.float32(3.402823466e+38)
    # reference scene: straight down onto the plane, onto the cube top, up from below (back faces culled)
    sc, osc = rtmi.reference_scene(), oracle.reference_scene()
    o = np.array([[1.5, 3, 1.5], [-1, 3, 0], [1.5, -3, 1.5], [-1, 1, 0]], dtype=np.float32)
    d = np.array([[0, -1, 0], [0, -1, 0], [0, 1, 0], [0, 1, 0]], dtype=np.float32)
    t, prim = _check_rays(rtmi, oracle, sc, osc, o, d, 1, 2)
    assert t[0] == 3.0 and prim[0] in (1, 2) and t[1] == 1.5 and prim[1] >= 3
    assert prim[2] == rtmi.abi.MISS_ID and prim[3] == rtmi.abi.MISS_ID   # from inside the cube every face is a back face


def test_closest_hit_random_rays(rtmi, oracle):
    rs = np.random.RandomState(5)
    n = 20000
    o = rs.uniform(-6, 6, size=(n, 3)).astype(np.float32)
    d = (rs.normal(size=(n, 3)) * rs.uniform(0.2, 3.0, size=(n, 1))).astype(np.float32)
    d[::7] = (np.array([0.0, 0.6, 0.0]) - o[::7]).astype(np.float32)          # aimed at the scene centre
    d[::11, 1] = 0.0                                                           # parallel to the ground slab
    case = dict(scene="spheres")
    _check_rays(rtmi, oracle, cases.product_scene(case), cases.oracle_scene(case), o, d, 11, 0)
    _check_rays(rtmi, oracle, rtmi.reference_scene(), oracle.reference_scene(), o, d, 1, 2)
    _check_rays(rtmi, oracle, rtmi.reference_scene(quad_rule=1), oracle.reference_scene(quad_rule=1), o, d, 0, 2)


def test_closest_hit_triangle_edges(rtmi, oracle):
    """Rays aimed exactly at triangle vertices, edge points and shared diagonals of the reference
    meshes, rays starting on the faces (t = 0 hits count), and grazing rays: the places where the
    kernels' early-out on the numerators of t, u, v has to agree with the shader's own test."""
    sc, osc = rtmi.reference_scene(), oracle.reference_scene()
    tris = np.array([[list(t.p1)[:3], list(t.p2)[:3], list(t.p3)[:3]] for t in list(osc.tris)[:14]], dtype=np.float32)
    rs = np.random.RandomState(21)
    targets = [tris[:, 0], tris[:, 1], tris[:, 2]]
    for _ in range(40):
        w = rs.dirichlet([0.3, 0.3, 0.3], size=14).astype(np.float32)   # many near edges and corners
        w[rs.rand(14) < 0.5, rs.randint(0, 3)] = 0.0                     # exactly on an edge
        w /= w.sum(axis=1, keepdims=True)
        targets.append((tris * w[:, :, None]).sum(axis=1))
    targets = np.concatenate(targets).astype(np.float32)
    origins = rs.uniform(-5, 5, size=targets.shape).astype(np.float32)
    origins[:, 1] = np.abs(origins[:, 1]) + 0.5
    dirs = (targets - origins) * rs.choice([0.25, 0.5, 1.0, 2.0], size=(len(targets), 1)).astype(np.float32)
    _check_rays(rtmi, oracle, sc, osc, origins, dirs, 1, 2)
    _check_rays(rtmi, oracle, sc, osc, origins, dirs, 0, 2)
    # rays that start on a face (what a bounce produces) and leave in random directions, some into the surface
    starts = np.repeat(targets[:200], 4, axis=0)
    out = rs.normal(size=starts.shape).astype(np.float32)
    _check_rays(rtmi, oracle, sc, osc, starts, out, 1, 2)
    # grazing along the plane and along cube faces
    graze_o = np.array([[-3, 0, 0.3], [-3, 1e-7, 0.3], [-3, 0.5, 0.2], [-3, 1.5, 0.2], [-1.5, 3, 0.1]], dtype=np.float32)
    graze_d = np.array([[1, 0, 0], [1, -1e-8, 0], [1, 0, 0], [1, 0, 0], [0, -1, 0]], dtype=np.float32)
    _check_rays(rtmi, oracle, sc, osc, graze_o, graze_d, 1, 2)


def test_closest_hit_near_ties_and_threshold(rtmi, oracle):
    """Rays built to land where comparing numerators and comparing rounded quotients could differ:
    nearly coincident spheres (both index orders), exactly coincident spheres (lowest index wins),
    roots at the 0.0001 threshold, and directions far outside the normal range."""
    eps = np.float32(2.0 ** -20)
    table = [((0.0, 0.0, 0.0), 1.0), ((0.0, 0.0, float(eps)), 1.0), ((0.0, 0.0, 0.0), 1.0),
             ((0.0, 0.0, -float(eps)), float(np.float32(1.0) + eps)), ((3.0, 0.0, 0.0), 2.0 ** -10)]
    sc, osc = _product_sphere_scene(rtmi, table), _oracle_sphere_scene(oracle, table)
    rs = np.random.RandomState(8)
    n = 4000
    o = (rs.normal(size=(n, 3)) * 3.0).astype(np.float32)
    d = (-o + rs.normal(size=(n, 3)) * 0.3).astype(np.float32)
    d *= rs.choice([0.25, 0.5, 1.0, 2.0, 3.0], size=(n, 1)).astype(np.float32)
    for ns in (5, 4, 3, 2):
        t, prim = _check_rays(rtmi, oracle, sc, osc, o, d, ns, 0)
    # exactly coincident spheres 0 and 2: sphere 2 never wins
    assert not (prim == 2).any()
    # roots around the t > 0.0001 threshold on the small sphere (index 4): origins one float apart
    r = np.float32(2.0 ** -10)
    steps = np.arange(-3000, 3000, dtype=np.int64)
    for scale in (1.0, 0.5, 3.0):
        # t is a ray parameter: the surface is 0.0001 * |d| away when t = 0.0001
        base = np.float32(r + np.float32(0.0001 * scale))
        z = (base.view(np.uint32).astype(np.int64) + steps).astype(np.uint32).view(np.float32)
        o2 = np.stack([np.full_like(z, 3.0), np.zeros_like(z), -z], axis=1)
        d2 = np.tile(np.array([0.0, 0.0, scale], dtype=np.float32), (len(z), 1))
        t, prim = _check_rays(rtmi, oracle, sc, osc, o2, d2, 5, 0)
        hit = prim == 4
        assert hit.any() and (~hit).any()          # the scan straddles the threshold
    # directions with denominators outside 2^-40 .. 2^40, zero, infinite and NaN directions
    o3 = np.tile(np.array([[0.0, 0.0, -3.0]], dtype=np.float32), (8, 1))
    d3 = np.array([[0, 0, 1e-25], [0, 0, 1e25], [0, 0, 0], [0, 0, np.inf], [np.nan, 0, 1], [0, 0, 1e-12], [0, 0, 3e19],
                   [1e-30, 1e-30, 1e-22]], dtype=np.float32)
    _check_rays(rtmi, oracle, sc, osc, o3, d3, 5, 0)
    _check_rays(rtmi, oracle, rtmi.reference_scene(), oracle.reference_scene(), o3, d3, 1, 2)
    # origins exactly on surfaces (what a bounce produces)
    o4 = np.array([[0, 0, -1], [0, 1, 0], [1, 0, 0], [3, 0, -(2.0 ** -10)]], dtype=np.float32)
    d4 = np.array([[0, 0, 1], [0, -1, 0.1], [-1, 0.2, 0], [0, 0, 1]], dtype=np.float32)
    _check_rays(rtmi, oracle, sc, osc, o4, d4, 5, 0)


# ---- checkpoint / resume -----------------------------------------------------------------------

@pytest.mark.parametrize("variant", VARIANTS)
def test_checkpoint_resume_is_bit_exact(rtmi, tmp_path, variant):
    """8 frames in one context == 3 frames, snapshot, new context, reload, 5 more frames."""
    case = dict(cases.CASES["c3_mesh_bench_pose"])
    sc = cases.product_scene(case)
    fd = cases.product_frame_data(case, sc)
    w, h = case["size"]
    path = tmp_path / "acc.snap"
    with rtmi.Renderer(w, h, variant=variant) as r:
        r.upload_scene(sc)
        r.render_frames(fd, 1, 8)
        want = (r.read_accum(), r.read_output_rgba8())
    with rtmi.Renderer(w, h, variant=variant) as r:
        r.upload_scene(sc)
        r.render_frames(fd, 1, 3)
        r.save_checkpoint(path, 3)
        mid = (r.read_accum(), r.read_output_rgba8())
    with rtmi.Renderer(w, h, variant=variant) as r:
        r.upload_scene(sc)
        assert r.load_checkpoint(path) == 3
        # the reloaded context shows what the interrupted one showed
        assert cases.bits_equal(r.read_accum(), mid[0]) and np.array_equal(r.read_output_rgba8(), mid[1])
        r.render_frames(fd, 4, 5)
        got = (r.read_accum(), r.read_output_rgba8())
    assert cases.bits_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_checkpoint_errors(rtmi, tmp_path):
    path = tmp_path / "acc.snap"
    with rtmi.Renderer(16, 8) as r:
        r.save_checkpoint(path, 0)
        assert r.load_checkpoint(path) == 0
        assert not r.read_output_rgba8().any()
        with pytest.raises(rtmi.RtmiError) as e:
            r.load_checkpoint(tmp_path / "missing.snap")
        assert e.value.code == rtmi.abi.ERR_IO
        with pytest.raises(rtmi.RtmiError) as e:
            r.save_checkpoint(tmp_path / "no_such_dir" / "x.snap", 1)
        assert e.value.code == rtmi.abi.ERR_IO
    with rtmi.Renderer(8, 16) as r:      # same pixel count, other extent
        with pytest.raises(rtmi.RtmiError) as e:
            r.load_checkpoint(path)
        assert e.value.code == rtmi.abi.ERR_INVALID_ARGUMENT
    data = path.read_bytes()
    for bad in (data[:-4], data + b"\0\0\0\0", b"RTMIACC2" + data[8:], data[:12]):
        path.write_bytes(bad)
        with rtmi.Renderer(16, 8) as r:
            with pytest.raises(rtmi.RtmiError) as e:
                r.load_checkpoint(path)
            assert e.value.code == rtmi.abi.ERR_IO


# ---- scripted camera: motion resets the accumulation, stillness continues it ------------------------

@pytest.mark.parametrize("variant", VARIANTS)
def test_scripted_camera_path_against_oracle(rtmi, oracle, variant):
    w, h = 64, 36
    sc, osc = rtmi.reference_scene(), oracle.reference_scene()
    fd = rtmi.make_frame_data(w, h, sc, rtmi.BENCH_POSE, rays_per_pixel=2, max_bounce=5)
    cam = rtmi.make_camera(w, h, rtmi.BENCH_POSE)
    D = rtmi.driver
    still = D.camera_input()
    steps = [(3, 1 / 60, still, False),                                            # frames 1, 2, 3
             (2, 0.25, D.camera_input("WDF", mouse=(40.0, -25.0), button2=True), False),   # moving: 1, 1
             (3, 1 / 60, D.camera_input(mouse=(40.0, -25.0)), False),               # 2, 3, 4
             (1, 1 / 60, D.camera_input(mouse=(40.0, -25.0)), True),                # reset key: 1
             (2, 1 / 60, D.camera_input("^", mouse=(40.0, -25.0)), False),          # moving up: 1, 1
             (2, 1 / 60, D.camera_input(mouse=(40.0, -25.0)), False)]               # 2, 3
    with rtmi.Renderer(w, h, variant=variant) as r:
        r.upload_scene(sc)
        log = D.replay(r, fd, cam, steps)
        got = dict(accum=r.read_accum(), rgba8=r.read_output_rgba8())
    assert [f for f, *_ in log] == [1, 2, 3, 1, 1, 2, 3, 4, 1, 1, 1, 2, 3]
    assert log[4][1] != log[2][1] and log[4][2] != log[2][2]      # the camera did move and turn
    accum = None
    for frame_index, pos, yaw, pitch in log:
        ofd = oracle.frame_data(w, h, pos=pos, yaw=yaw, pitch=pitch, rpp=2, max_bounce=5, scene=osc)
        want = oracle.render(ofd, osc, frame_index, 1, accum=accum, want_debug=False)
        accum = want["accum"]
    same = cases.bits_equal(got["accum"], want["accum"])
    assert same, "accumulation differs in %d pixels" % int((got["accum"] != want["accum"]).any(axis=-1).sum())
    assert np.array_equal(got["rgba8"], want["rgba8"])


def test_short_exact_routines_match_for_every_admitted_input(rtmi):
    """The kernels' short sqrt / reciprocal / division-by-constant sequences against the general
    correctly rounded operations, for every binary32 bit pattern they are used for."""
    with rtmi.Renderer(8, 8) as r:
        for op, divisor, at_least in ((0, 1.0, 0x70000000), (1, 1.0, 0x3C000000),
                                      (2, float(np.float32(0.4)), 0x78000000), (2, float(np.float32(0.01)), 0x78000000)):
            tested, bad, first = r.selftest_exhaustive(op, divisor)
            assert tested >= at_least, (op, divisor, tested)
            assert bad == 0, "op %d divisor %g: %d mismatches, first at bit pattern 0x%08x" % (op, divisor, bad, first)
