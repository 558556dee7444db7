"""Hierarchy path with more Mesh records than one 64-bit gate mask holds (the walk then moves from one
chunk of 64 gates to the next per lane), checked against the oracle's exhaustive loops."""
import numpy as np
import pytest

from test_gpu_bvh import _assert_rays_equal, _hits, _random_rays

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
