"""GPU tests of the bounding-volume-hierarchy path (run on an MI355X with `-m gpu`).

The hierarchy must not change any result: every test compares it bit for bit with the exhaustive
loops, either the CPU oracle's (Raytracing.comp:338-388 restated) or the GPU's own RTMI_ACCEL_NONE
kernels, on scenes of about 10^4 primitives, where the exhaustive loops are still affordable."""
import numpy as np
import pytest

import cases
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


def _small_stress(rtmi, **kw):
    args = dict(n_spheres=5000, grid=(50, 50), n_meshes=4, seed=12345, extent=5.0)
    args.update(kw)
    return rtmi.scenes.stress_scene(**args)


def _hits(rtmi, sc, accel, o, d, ns=None, nm=None):
    with rtmi.Renderer(8, 8) as r:
        r.set_acceleration(accel)
        r.upload_scene(sc)
        assert r.acceleration() == (rtmi.abi.ACCEL_BVH if accel == rtmi.abi.ACCEL_BVH else rtmi.abi.ACCEL_NONE)
        return r.selftest_closest_hit(sc.n_spheres if ns is None else ns, sc.n_meshes if nm is None else nm, o, d)


def _assert_rays_equal(got, want, what):
    t, prim = got
    wt, wprim = want
    same_p = np.array_equal(prim, wprim)
    assert same_p, "%s: prim ids differ for %d of %d rays" % (what, int((prim != wprim).sum()), len(prim))
    same_t = cases.bits_equal(t, wt)
    assert same_t, "%s: t differs for %d rays" % (what, int((t.view(np.uint32) != wt.view(np.uint32)).sum()))


def _random_rays(seed, n, extent, far=1.0):
    rs = np.random.RandomState(seed)
    o = rs.uniform(-1.3 * extent * far, 1.3 * extent * far, size=(n, 3)).astype(np.float32)
    d = (rs.normal(size=(n, 3)) * rs.uniform(0.2, 3.0, size=(n, 1))).astype(np.float32)
    target = rs.uniform(-extent, extent, size=(n, 3))
    aimed = np.arange(n) % 3 != 0                    # two rays in three are aimed into the scene
    d[aimed] = (target[aimed] - o[aimed]).astype(np.float32) * rs.uniform(0.01, 2.0, size=(int(aimed.sum()), 1)).astype(np.float32)
    d[::13, 0] = 0.0                                 # axis-parallel components: 1/0 in the slab tests
    d[::17, 1] = 0.0
    d[::19, 2] = 0.0
    return o, d


def test_auto_mode_threshold(rtmi):
    small = rtmi.sphere_scene()
    big, _ = _small_stress(rtmi, n_spheres=500, grid=(0, 0), n_meshes=0)
    with rtmi.Renderer(8, 8) as r:
        r.upload_scene(small)
        assert r.acceleration() == rtmi.abi.ACCEL_NONE
        r.upload_scene(big)
        assert r.acceleration() == rtmi.abi.ACCEL_BVH
        r.set_acceleration(rtmi.abi.ACCEL_NONE)
        r.upload_scene(big)
        assert r.acceleration() == rtmi.abi.ACCEL_NONE
        r.set_acceleration(rtmi.abi.ACCEL_BVH)
        r.upload_scene(small)
        assert r.acceleration() == rtmi.abi.ACCEL_BVH
        with pytest.raises(rtmi.RtmiError):
            r.set_acceleration(7)


def test_closest_hit_matches_oracle_and_exhaustive_kernels(rtmi, oracle):
    sc, _ = _small_stress(rtmi)
    osc = oracle.Scene.from_product(sc)
    o, d = _random_rays(1, 30000, 5.0)
    bvh = _hits(rtmi, sc, rtmi.abi.ACCEL_BVH, o, d)
    assert int((bvh[1] != rtmi.abi.MISS_ID).sum()) > 5000       # the rays do find things
    assert int((bvh[1] >= sc.n_spheres).sum()) > 500 and int((bvh[1] < sc.n_spheres).sum()) > 500
    _assert_rays_equal(bvh, _hits(rtmi, sc, rtmi.abi.ACCEL_NONE, o, d), "bvh vs exhaustive kernels")
    _assert_rays_equal(bvh, oracle.closest_hit(osc, o, d), "bvh vs oracle")


def test_closest_hit_far_origins(rtmi, oracle):
    """Rays that start hundreds of scene sizes away: the tests' rounding errors are then far larger
    than the spheres, and the box margins have to grow with the distance."""
    sc, _ = _small_stress(rtmi)
    osc = oracle.Scene.from_product(sc)
    for far in (20.0, 1000.0):
        o, d = _random_rays(2, 20000, 5.0, far=far)
        bvh = _hits(rtmi, sc, rtmi.abi.ACCEL_BVH, o, d)
        _assert_rays_equal(bvh, oracle.closest_hit(osc, o, d), "far=%g" % far)


def test_closest_hit_subset_of_the_uploaded_scene(rtmi, oracle):
    """Frame data may name fewer spheres and meshes than were uploaded (Raytracing.comp:343, 352)."""
    sc, _ = _small_stress(rtmi)
    osc = oracle.Scene.from_product(sc)
    o, d = _random_rays(3, 10000, 5.0)
    for ns, nm in ((1234, 2), (0, 4), (5000, 0), (1, 1)):
        bvh = _hits(rtmi, sc, rtmi.abi.ACCEL_BVH, o, d, ns, nm)
        _assert_rays_equal(bvh, oracle.closest_hit(osc, o, d, ns, nm), "subset %d/%d" % (ns, nm))


def test_ties_keep_the_loop_order(rtmi, oracle):
    """Coincident primitives give equal t; the shader's strict `<` keeps the first one it visits."""
    base, _ = _small_stress(rtmi, n_spheres=600, grid=(12, 12), n_meshes=3)
    ns, nt, nm = base.n_spheres, base.n_triangles, base.n_meshes
    sph = np.frombuffer(base.sphere_array(), dtype=rtmi.host.SPHERE_DTYPE, count=ns)
    tri = np.frombuffer(base.triangles, dtype=rtmi.host.TRIANGLE_DTYPE, count=nt)
    msh = np.frombuffer(base.meshes, dtype=rtmi.host.MESH_DTYPE, count=nm)
    # every sphere three times (reversed in the middle), every triangle twice within its mesh, and
    # the meshes' ranges listed again in reverse order as further meshes over the same triangles
    sph3 = np.concatenate([sph, sph[::-1], sph])
    tri2 = np.empty(2 * nt, dtype=tri.dtype)
    msh2 = np.concatenate([msh, msh[::-1]]).copy()
    pos = 0
    for m in range(nm):
        s, n = int(msh["start_triangle"][m]), int(msh["num_triangles"][m])
        tri2[pos:pos + n] = tri[s:s + n]
        tri2[pos + n:pos + 2 * n] = tri[s:s + n][::-1]
        msh2["start_triangle"][m], msh2["num_triangles"][m] = pos, 2 * n
        msh2["start_triangle"][2 * nm - 1 - m], msh2["num_triangles"][2 * nm - 1 - m] = pos, 2 * n
        pos += 2 * n
    sc = rtmi.Scene.from_records(sph3, tri2, msh2)
    osc = oracle.Scene.from_product(sc)
    o, d = _random_rays(4, 20000, 5.0)
    bvh = _hits(rtmi, sc, rtmi.abi.ACCEL_BVH, o, d)
    want = oracle.closest_hit(osc, o, d)
    _assert_rays_equal(bvh, want, "duplicated primitives")
    hit_s = bvh[1] < sc.n_spheres
    assert hit_s.sum() > 300 and (bvh[1][hit_s] < ns).all()    # always the first copy
    _assert_rays_equal(bvh, _hits(rtmi, sc, rtmi.abi.ACCEL_NONE, o, d), "duplicated primitives, exhaustive kernels")


@pytest.mark.parametrize("variant", [0, 1, 2, 4])
def test_render_matches_oracle(rtmi, oracle, variant):
    sc, pose = _small_stress(rtmi)
    osc = oracle.Scene.from_product(sc)
    w, h, first, n = 64, 36, 1, 2
    fd = rtmi.make_frame_data(w, h, sc, pose=pose, rays_per_pixel=2, max_bounce=16)
    ofd = oracle.frame_data(w, h, pos=pose[0], yaw=pose[1], pitch=pose[2], rpp=2, max_bounce=16, scene=osc)
    want = oracle.render(ofd, osc, first, n)
    with rtmi.Renderer(w, h, variant=variant) as r:
        r.set_acceleration(rtmi.abi.ACCEL_BVH)
        r.upload_scene(sc)
        r.enable_validation(True)
        r.render_frames(fd, first, n)
        got = dict(accum=r.read_accum(), rgba8=r.read_output_rgba8(), prim_id=r.read_hit_ids(),
                   chain_hash=r.read_chain_hash(), rng_state=r.read_rng_state())
        cn = r.counters()
    assert_same(got, want)
    hit_px = int((want["prim_id"] != oracle.MISS_ID).sum())
    assert hit_px > w * h // 2
    # events that do not depend on how candidates are found agree with the oracle's counts; the
    # test counts are those the hierarchy actually made, far fewer than the exhaustive loops'
    for k in ("n_sphere_wins", "n_tri_wins", "n_hit_bounces", "n_env_evals", "n_pixel_frames", "n_paths", "n_segments"):
        assert cn[k] == want["counters"][k], k
    assert 0 < cn["n_sphere_tests"] < want["counters"]["n_sphere_tests"] // 20
    assert 0 < cn["n_tri_tests"] < want["counters"]["n_tri_tests"] // 20


def test_stream_variant_matches_compact_across_launches(rtmi):
    """The pixel-streaming kernel against the compacting one, without validation, on an image whose
    sides are not multiples of the 8 x 8 hand-out tile, over launches that continue a running sum."""
    sc, pose = _small_stress(rtmi, seed=4242)
    w, h = 100, 52
    fd = rtmi.make_frame_data(w, h, sc, pose=pose, rays_per_pixel=3, max_bounce=16)
    out = {}
    for variant in (rtmi.abi.VARIANT_STREAM, rtmi.abi.VARIANT_COMPACT):
        with rtmi.Renderer(w, h, variant=variant) as r:
            r.set_acceleration(rtmi.abi.ACCEL_BVH)
            r.upload_scene(sc)
            r.render_frames(fd, 1, 1)
            first = r.read_accum()
            r.render_frames(fd, 2, 3)
            out[variant] = (first, r.read_accum(), r.read_output_rgba8())
    a, b = out[rtmi.abi.VARIANT_STREAM], out[rtmi.abi.VARIANT_COMPACT]
    assert cases.bits_equal(a[0], b[0]) and cases.bits_equal(a[1], b[1]) and (a[2] == b[2]).all()
    assert (a[1][..., 3] == 1.0).all() and np.abs(a[1][..., :3]).sum() > 0


def test_render_matches_exhaustive_kernels_without_validation(rtmi):
    sc, pose = _small_stress(rtmi, seed=777)
    w, h = 96, 54
    fd = rtmi.make_frame_data(w, h, sc, pose=pose, rays_per_pixel=1, max_bounce=16)
    out = []
    for accel in (rtmi.abi.ACCEL_BVH, rtmi.abi.ACCEL_NONE):
        with rtmi.Renderer(w, h) as r:
            r.set_acceleration(accel)
            r.upload_scene(sc)
            r.render_frames(fd, 1, 3)
            out.append((r.read_accum(), r.read_output_rgba8()))
    same = cases.bits_equal(out[0][0], out[1][0])
    assert same, "accumulation images differ in %d pixels" % int((out[0][0] != out[1][0]).any(axis=-1).sum())
    assert np.array_equal(out[0][1], out[1][1])


def test_degenerate_scenes(rtmi, oracle):
    """One primitive, empty meshes, and a sphere list with NaN / infinite / zero-radius entries."""
    sph = np.zeros(6, dtype=rtmi.host.SPHERE_DTYPE)
    sph["center"] = [[0, 0, 0], [np.nan, 0, 0], [2, 0, 0], [0, 3, 0], [0, 0, 0], [4, 4, 4]]
    sph["radius"] = [1.0, 1.0, np.inf, 0.0, -0.5, np.nan]
    tri = np.zeros(1, dtype=rtmi.host.TRIANGLE_DTYPE)
    tri["p1"][0, :3], tri["p2"][0, :3], tri["p3"][0, :3] = (-1, -2, -1), (-1, -2, 1), (1, -2, 1)
    msh = np.zeros(2, dtype=rtmi.host.MESH_DTYPE)
    msh["num_triangles"] = [1, 0]
    msh["bounding_point1"][:, :3] = (-1, -2, -1)
    msh["bounding_point2"][:, :3] = (1, -2, 1)
    sc = rtmi.Scene.from_records(sph, tri, msh)
    osc = oracle.Scene.from_product(sc)
    o, d = _random_rays(6, 5000, 3.0)
    for ns, nm in ((6, 2), (1, 1), (2, 0), (0, 2)):
        bvh = _hits(rtmi, sc, rtmi.abi.ACCEL_BVH, o, d, ns, nm)
        _assert_rays_equal(bvh, oracle.closest_hit(osc, o, d, ns, nm), "degenerate %d/%d" % (ns, nm))


def test_full_size_stress_scene(rtmi, oracle):
    """The 10^6-sphere + 10^6-triangle scene itself: closest hits of random rays against the oracle's
    exhaustive loops (a few thousand rays are affordable), and two properties that need no oracle: the
    nested-loop kernel (one uninterrupted walk per segment) and the wave-cooperative and pixel-streaming
    kernels (walks stopped and resumed) give the same image, and so does splitting the frames over two
    launches.  At 1920 x 1080 the two resuming kernels are compared with each other."""
    sc, pose = rtmi.scenes.stress_scene()
    assert (sc.n_spheres, sc.n_triangles, sc.n_meshes) == (1000000, 1000000, 16)
    osc = oracle.Scene.from_product(sc)
    o, d = _random_rays(11, 3000, 50.0)
    w, h = 96, 54
    fd = rtmi.make_frame_data(w, h, sc, pose=pose, rays_per_pixel=1, max_bounce=16)
    out = {}
    for variant in (0, 2, 4):
        with rtmi.Renderer(w, h, variant=variant) as r:
            r.upload_scene(sc)
            assert r.acceleration() == rtmi.abi.ACCEL_BVH
            if variant == 2:
                rays = r.selftest_closest_hit(sc.n_spheres, sc.n_meshes, o, d)
            r.enable_validation(True)
            r.render_frames(fd, 1, 4)
            out[variant] = dict(accum=r.read_accum(), rgba8=r.read_output_rgba8(), prim_id=r.read_hit_ids(),
                                chain_hash=r.read_chain_hash(), rng_state=r.read_rng_state())
            if variant == 2:
                r.render_frames(fd, 1, 1)
                r.render_frames(fd, 2, 3)
                out["split"] = dict(accum=r.read_accum(), rgba8=r.read_output_rgba8())
    assert_same(out[2], out[0])
    assert_same(out[4], out[0])
    assert_same(out["split"], out[2], keys=("rgba8",))
    assert int((out[2]["prim_id"] != rtmi.abi.MISS_ID).sum()) > w * h // 2
    assert int((rays[1] != rtmi.abi.MISS_ID).sum()) > 500
    _assert_rays_equal(rays, oracle.closest_hit(osc, o, d), "full stress scene")
    # two samples per pixel, uninstrumented kernels: the streaming kernel's build for large scenes with a sample loop
    fd2 = rtmi.make_frame_data(w, h, sc, pose=pose, rays_per_pixel=2, max_bounce=16)
    two = []
    for variant in (0, 4):
        with rtmi.Renderer(w, h, variant=variant) as r:
            r.upload_scene(sc)
            r.render_frames(fd2, 1, 2)
            two.append(r.read_accum())
    assert cases.bits_equal(two[0], two[1])
    big = {}
    fd = rtmi.make_frame_data(1920, 1080, sc, pose=pose, rays_per_pixel=1, max_bounce=16)
    for variant in (2, 4):
        with rtmi.Renderer(1920, 1080, variant=variant) as r:
            r.upload_scene(sc)
            r.render_frames(fd, 1, 2)
            big[variant] = (r.read_accum(), r.read_output_rgba8())
    assert cases.bits_equal(big[2][0], big[4][0]) and np.array_equal(big[2][1], big[4][1])


def test_walk_only_streaming_kernel_returns_the_same_hits(rtmi):
    """rtmi_selftest_walk (the measurement hook that walks rays through persistent waves with resumable walks, as
    the path kernels do) against rtmi_selftest_closest_hit, ray by ray; more rays than one round of the device's
    wave slots holds, and a count that is not a multiple of the wave size."""
    sc, _ = rtmi.scenes.stress_scene(4000, (40, 40), 3, seed=8, extent=4.0)
    rs = np.random.RandomState(12)
    n = 64 * 6144 + 37
    o = rs.uniform(-5, 5, size=(n, 3)).astype(np.float32)
    d = (rs.normal(size=(n, 3)) * rs.uniform(0.3, 1.4, size=(n, 1))).astype(np.float32)
    with rtmi.Renderer(8, 8) as r:
        r.set_acceleration(rtmi.abi.ACCEL_BVH)
        r.upload_scene(sc)
        t0, q0 = r.selftest_closest_hit(sc.n_spheres, sc.n_meshes, o, d)
        t1, q1, ms = r.selftest_walk(sc.n_spheres, sc.n_meshes, o, d, repeats=2)
    assert (q0 != rtmi.abi.MISS_ID).mean() > 0.3 and ms > 0.0
    assert np.array_equal(q0, q1) and np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
