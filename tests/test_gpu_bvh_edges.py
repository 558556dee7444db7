"""Hierarchy path at the corners of the frame loop: more frames than the kernel's frame ring holds,
bounce limits 0 / 1 / 2 (where the launcher falls back to the simpler kernels), several samples per
pixel, a continued accumulation, ragged image sizes.  Everything against the oracle's exhaustive loops."""
import numpy as np
import pytest

import cases
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


def _render_both(rtmi, oracle, sc, osc, pose, w, h, rpp, max_bounce, first, n, variant, seed_accum=None):
    fd = rtmi.make_frame_data(w, h, sc, pose=pose, rays_per_pixel=rpp, max_bounce=max_bounce)
    ofd = oracle.frame_data(w, h, pos=pose[0], yaw=pose[1], pitch=pose[2], rpp=rpp, max_bounce=max_bounce, scene=osc)
    want = oracle.render(ofd, osc, first, n, accum=seed_accum)
    with rtmi.Renderer(w, h, variant=variant) as r:
        r.set_acceleration(rtmi.abi.ACCEL_BVH)
        r.upload_scene(sc)
        r.enable_validation(True)
        if seed_accum is not None:
            r.write_accum(seed_accum)
        r.render_frames(fd, first, n)
        got = dict(accum=r.read_accum(), rgba8=r.read_output_rgba8(), prim_id=r.read_hit_ids(),
                   chain_hash=r.read_chain_hash(), rng_state=r.read_rng_state())
    return got, want


@pytest.fixture(scope="module")
def scene(rtmi, oracle):
    sc, pose = rtmi.scenes.stress_scene(2500, (30, 30), 3, seed=5, extent=4.0)
    return sc, oracle.Scene.from_product(sc), pose


@pytest.mark.parametrize("variant", [0, 1, 2, 4])
def test_more_frames_than_the_ring(rtmi, oracle, scene, variant):
    sc, osc, pose = scene
    got, want = _render_both(rtmi, oracle, sc, osc, pose, 48, 27, 1, 12, 1, 11, variant)
    assert_same(got, want)


@pytest.mark.parametrize("variant", [2, 4])
@pytest.mark.parametrize("max_bounce", [0, 1, 2, 3])
def test_small_bounce_limits(rtmi, oracle, scene, max_bounce, variant):
    sc, osc, pose = scene
    got, want = _render_both(rtmi, oracle, sc, osc, pose, 40, 24, 2, max_bounce, 1, 3, variant)
    assert_same(got, want)


@pytest.mark.parametrize("variant", [2, 4])
def test_samples_per_pixel_and_continued_sum(rtmi, oracle, scene, variant):
    sc, osc, pose = scene
    w, h = 37, 19                                   # not a multiple of the tile or the wave
    rs = np.random.RandomState(2)
    seed = rs.uniform(0.0, 4.0, size=(h, w, 4)).astype(np.float32)
    seed[..., 3] = 1.0
    got, want = _render_both(rtmi, oracle, sc, osc, pose, w, h, 3, 7, 5, 6, variant, seed_accum=seed)
    assert_same(got, want)


def test_frame_loop_equals_batched_launch(rtmi, scene):
    sc, _, pose = scene
    w, h = 64, 36
    fd = rtmi.make_frame_data(w, h, sc, pose=pose, rays_per_pixel=1, max_bounce=16)
    with rtmi.Renderer(w, h) as r:
        r.set_acceleration(rtmi.abi.ACCEL_BVH)
        r.upload_scene(sc)
        r.render_frames(fd, 1, 9)
        batched = (r.read_accum(), r.read_output_rgba8())
        for f in range(1, 10):
            fd.frame_index = f
            r.render(fd)
        looped = (r.read_accum(), r.read_output_rgba8())
    assert cases.bits_equal(batched[0], looped[0]) and np.array_equal(batched[1], looped[1])


@pytest.mark.parametrize("size", [(1, 1), (7, 3), (9, 17)])
def test_images_smaller_than_a_tile(rtmi, oracle, scene, size):
    """The streaming kernel hands pixels out in 8 x 8 tiles; images that do not fill one, or end in
    partial ones, must come out like any other."""
    sc, osc, pose = scene
    got, want = _render_both(rtmi, oracle, sc, osc, pose, size[0], size[1], 2, 6, 1, 3, 4)
    assert_same(got, want)


def _closest_hit_both_ways(rtmi, sc, o32, d32, cap=None):
    res = {}
    for mode in (rtmi.abi.ACCEL_NONE, rtmi.abi.ACCEL_BVH):
        with rtmi.Renderer(8, 8) as r:
            r.set_acceleration(mode)
            if cap is not None:
                r.set_grazing_cap(cap)
            r.upload_scene(sc)
            assert r.acceleration() == mode
            res[mode] = r.selftest_closest_hit(sc.n_spheres, sc.n_meshes, o32, d32)
    (t0, q0), (t1, q1) = res[rtmi.abi.ACCEL_NONE], res[rtmi.abi.ACCEL_BVH]
    differ = (q0 != q1) | (t0.view(np.uint32) != t1.view(np.uint32))
    return differ, q0, q1


def test_rays_grazing_large_triangles(rtmi):
    """Rays that run almost inside the plane of a large triangle and pass just outside one of its edges.

    The triangle test (Raytracing.comp:286-309) carries a rounding error of about 8 ulp of |ao| / cos(phi) in
    the plane (phi = angle between ray and normal), so within a fraction of a degree of the plane it accepts
    rays that geometrically miss, by more than the 2^-12 |ao| by which the hierarchy grows its boxes (with the
    margins of the first version 1 of these 60 000 rays differed).  The hierarchy visits the nodes above a
    large triangle unconditionally for rays inside the band around its plane: no ray may differ."""
    rs = np.random.RandomState(17)
    n_tri = 400
    f = np.float32
    centre = rs.uniform(-20, 20, size=(n_tri, 3))
    e1 = rs.normal(size=(n_tri, 3))
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    tmp = rs.normal(size=(n_tri, 3))
    e2 = np.cross(e1, tmp)
    e2 /= np.linalg.norm(e2, axis=1, keepdims=True)
    size = rs.uniform(2.0, 10.0, size=(n_tri, 1))
    p1 = centre - 0.5 * size * e1 - 0.3 * size * e2
    p2 = centre + 0.5 * size * e1 - 0.3 * size * e2
    p3 = centre + 0.6 * size * e2
    tris = np.zeros(n_tri, dtype=rtmi.host.TRIANGLE_DTYPE)
    for name, v in (("p1", p1), ("p2", p2), ("p3", p3)):
        tris[name][:, :3] = v.astype(f)
    nrm = np.cross(p2 - p1, p3 - p1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    for name in ("n1", "n2", "n3"):
        tris[name][:, :3] = nrm.astype(f)
    lo = np.minimum(np.minimum(p1, p2), p3).min(axis=0) - 1.0
    hi = np.maximum(np.maximum(p1, p2), p3).max(axis=0) + 1.0
    mesh = np.zeros(1, dtype=rtmi.host.MESH_DTYPE)
    mesh["start_triangle"], mesh["num_triangles"] = 0, n_tri
    mesh["bounding_point1"][0, :3], mesh["bounding_point2"][0, :3] = lo.astype(f), hi.astype(f)
    mesh["material"]["color"][0] = (1.0, 1.0, 1.0)
    sc = rtmi.Scene.from_records(np.zeros(0, dtype=rtmi.host.SPHERE_DTYPE), tris, mesh)
    # rays: towards a point just off an edge (outside or inside by a relative delta), inside the plane up to a tiny tilt
    n_rays = 60000
    k = rs.randint(0, n_tri, n_rays)
    a, b = p1[k], p2[k]                                   # the edge p1 -> p2; outward is -e2
    s = rs.uniform(0.05, 0.95, size=(n_rays, 1))
    delta = 10.0 ** rs.uniform(-7, -1.5, size=(n_rays, 1)) * size[k] * rs.choice([-1.0, 1.0], size=(n_rays, 1))
    target = a + s * (b - a) - delta * e2[k]
    along = rs.normal(size=(n_rays, 1)) * e1[k] + rs.normal(size=(n_rays, 1)) * e2[k]
    along /= np.linalg.norm(along, axis=1, keepdims=True)
    tilt = 10.0 ** rs.uniform(-7, -2, size=(n_rays, 1))   # radians off the plane, towards the front side
    direction = along - tilt * nrm[k]
    origin = target - direction * rs.uniform(1.0, 30.0, size=(n_rays, 1))
    o32, d32 = origin.astype(f), (direction * rs.choice([0.5, 1.0, 2.0], size=(n_rays, 1))).astype(f)
    differ, q0, _ = _closest_hit_both_ways(rtmi, sc, o32, d32)
    assert (q0 != rtmi.abi.MISS_ID).mean() > 0.2           # the probe does hit triangles
    assert differ.sum() == 0, "%d of %d rays differ" % (int(differ.sum()), n_rays)


@pytest.mark.parametrize("cap", [None, 0.0])
def test_rays_along_triangle_planes_of_every_size(rtmi, cap):
    """The same question on four meshes (cells of the stress scene's grid with white-noise heights; triangles of
    2-10, 0.3-2 and 0.05-0.4 units in random orientation) and on a mesh that mixes the sizes, 40 000 rays each
    with tilts of 1e-7 ... 1e-2 radians off a triangle's plane, aimed at points up to a few edge lengths away
    from the triangle: hierarchy and exhaustive loops must agree on every one, with the default cap on the
    margins of small triangles and without any (large triangles do not depend on the cap)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import grazing_rate as G
    rs = np.random.RandomState(23)
    big, small = G.big_triangles(rs, 60), G.big_triangles(rs, 3000, (0.05, 0.4))
    for name, tri in (("noisy_grid", G.noisy_grid(rs)), ("big", G.big_triangles(rs)), ("mid", G.big_triangles(rs, 2000, (0.3, 2.0))),
                      ("small", G.big_triangles(rs, 8000, (0.05, 0.4))), ("mixed", np.concatenate([small[:1500], big, small[1500:]]))):
        sc = G.one_mesh_scene(tri)
        o, d = G.rays_grazing(rs, tri, 40000)
        differ, q0, _ = _closest_hit_both_ways(rtmi, sc, o.astype(np.float32), d.astype(np.float32), cap=cap)
        assert (q0 != rtmi.abi.MISS_ID).mean() > 0.02, name
        assert differ.sum() == 0, "%s: %d rays differ" % (name, int(differ.sum()))
