"""The default variant keeps the primary segment of a static camera between launches.  These tests
drive the sequences in which that store is built, used and has to be dropped, and compare every
step with launches that never use it (explicit variant 1)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _both(rtmi, w, h):
    return rtmi.Renderer(w, h, variant=rtmi.abi.VARIANT_AUTO), rtmi.Renderer(w, h, variant=rtmi.abi.VARIANT_REGENERATE)


def _same(a, b):
    return cases.bits_equal(a.read_accum(), b.read_accum()) and np.array_equal(a.read_output_rgba8(), b.read_output_rgba8())


@pytest.mark.parametrize("w,h,cus", [(96, 54, 0), (192, 108, 1)])
def test_store_use_and_invalidation(rtmi, w, h, cus):
    """cus = 1: the scheduling decisions are made for a device of one compute unit, so that the launches of the
    compacting pair that start from the store take waves of neighbouring pixels, as large images do on the real device."""
    ref, sph = rtmi.reference_scene(), rtmi.sphere_scene()
    auto, plain = _both(rtmi, w, h)
    with auto, plain:
        for r in (auto, plain):
            r.assume_compute_units(cus)
            r.upload_scene(ref)
        if cus:
            assert auto.paths_pixels_per_wave(w * h // 4, 20, 2, 2)[1] is True
        fd = rtmi.make_frame_data(w, h, ref, rtmi.BENCH_POSE, rays_per_pixel=2, max_bounce=6)
        frame = 0

        def step(n=1, **change):
            nonlocal frame, fd
            for k, v in change.items():
                setattr(fd, k, v)
            for r in (auto, plain):
                r.render_frames(fd, frame + 1, n)
            frame += n
            assert _same(auto, plain), "after frame %d (%r)" % (frame, change)

        for _ in range(5):
            step()                                  # plain launch, storing launch, three launches from the store
        step(3)                                     # more frames in one launch, still from the store
        step(rays_per_pixel=3)                      # a frame constant changed: the store is stale
        step()
        step()
        step(sun_focus=100.0)                       # sky constants are part of what is stored
        step()
        step()
        fd2 = rtmi.make_frame_data(w, h, ref, ((0.5, 1.2, -4.0), 80.0, -10.0), rays_per_pixel=3, max_bounce=6)
        fd2.sun_focus = 100.0
        fd = fd2
        frame = 0                                   # the camera moved: the sum starts again
        for _ in range(4):
            step()
        for r in (auto, plain):                     # another scene under the same camera
            r.upload_scene(sph)
        fd.sphere_number, fd.mesh_number = sph.n_spheres, 0
        frame = 0
        for _ in range(4):
            step()
        step(20)                                    # a long launch goes to the compacting pair, which stores as it goes
        step(20)                                    # and starts from the store and its work list the next time
        step(8)
        step(4)                                     # a small image: the pair from four frames on
        step(3)                                     # fewer: the one-lane-per-pixel kernel, from the same store
        step(2)
        # an explicit variant-2 launch rebuilds the work list; the next default launch must not trust the store
        auto.set_variant(rtmi.abi.VARIANT_COMPACT)
        fd3 = rtmi.make_frame_data(w, h, sph, rtmi.DEFAULT_POSE, rays_per_pixel=3, max_bounce=6)
        auto.render_frames(fd3, 1, 1)
        plain.render_frames(fd3, 1, 1)
        auto.set_variant(rtmi.abi.VARIANT_AUTO)
        for r in (auto, plain):
            r.render_frames(fd, 1, 1)
        assert _same(auto, plain)
        frame = 1
        for _ in range(3):
            step()


def test_store_with_a_hierarchy(rtmi):
    w, h = 64, 36
    sc, pose = rtmi.scenes.stress_scene(1500, (20, 20), 3, seed=9, extent=4.0)
    fd = rtmi.make_frame_data(w, h, sc, pose, rays_per_pixel=1, max_bounce=10)
    auto = rtmi.Renderer(w, h, variant=rtmi.abi.VARIANT_AUTO)
    plain = rtmi.Renderer(w, h, variant=rtmi.abi.VARIANT_COMPACT)
    with auto, plain:
        for r in (auto, plain):
            r.upload_scene(sc)
            assert r.acceleration() == rtmi.abi.ACCEL_BVH
        for f in range(1, 7):
            for r in (auto, plain):
                r.render_frames(fd, f, 1)
            assert _same(auto, plain), "frame %d" % f
