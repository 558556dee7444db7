"""Hierarchy path with more Mesh records than one 64-bit gate mask holds (the walk then moves from one
chunk of 64 gates to the next per lane), checked against the oracle's exhaustive loops."""
import numpy as np
import pytest

from test_gpu_bvh import _assert_rays_equal, _hits, _random_rays
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_meshes", [64, 65, 150])
def test_more_than_64_meshes(rtmi, oracle, n_meshes):
    sc, pose = rtmi.scenes.stress_scene(1500, (16, 300), n_meshes, seed=21, extent=5.0)
    assert sc.n_meshes == n_meshes and sc.n_triangles == 9600
    osc = oracle.Scene.from_product(sc)
    o, d = _random_rays(8, 12000, 5.0)
    o[::2, 1] = np.abs(o[::2, 1]) + 1.0          # half of the rays start above the terrain
    d[::2, 1] = -np.abs(d[::2, 1]) - 0.05        # and head down into it
    bvh = _hits(rtmi, sc, rtmi.abi.ACCEL_BVH, o, d)
    assert int((bvh[1] >= sc.n_spheres).sum()) > 1000          # plenty of triangle hits
    _assert_rays_equal(bvh, oracle.closest_hit(osc, o, d), "%d meshes" % n_meshes)
    # frame data naming only some of the meshes, across the chunk boundary
    for nm in (63, n_meshes - 1):
        part = _hits(rtmi, sc, rtmi.abi.ACCEL_BVH, o, d, sc.n_spheres, nm)
        _assert_rays_equal(part, oracle.closest_hit(osc, o, d, sc.n_spheres, nm), "%d of %d meshes" % (nm, n_meshes))


@pytest.mark.parametrize("variant", [2, 4])
def test_render_with_150_meshes(rtmi, oracle, variant):
    """The resumable walks of the rendering kernels across gate chunks (the ray tests above go through
    the walk-to-the-end helper)."""
    sc, pose = rtmi.scenes.stress_scene(1500, (16, 300), 150, seed=21, extent=5.0)
    osc = oracle.Scene.from_product(sc)
    w, h = 40, 22
    fd = rtmi.make_frame_data(w, h, sc, pose=pose, rays_per_pixel=1, max_bounce=8)
    ofd = oracle.frame_data(w, h, pos=pose[0], yaw=pose[1], pitch=pose[2], rpp=1, max_bounce=8, scene=osc)
    want = oracle.render(ofd, osc, 1, 2)
    with rtmi.Renderer(w, h, variant=variant) as r:
        r.set_acceleration(rtmi.abi.ACCEL_BVH)
        r.upload_scene(sc)
        r.enable_validation(True)
        r.render_frames(fd, 1, 2)
        got = dict(accum=r.read_accum(), rgba8=r.read_output_rgba8(), prim_id=r.read_hit_ids(),
                   chain_hash=r.read_chain_hash(), rng_state=r.read_rng_state())
    assert_same(got, want)
    assert int((want["prim_id"] >= sc.n_spheres).sum()) > 50      # triangles are seen
