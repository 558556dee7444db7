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
