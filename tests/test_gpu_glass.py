"""The transmitting-material extension (rtmi_enable_transmission).  The reference has no such material
(Raytracing.comp:26-34), so there is no parity target: what is checked is that the kernels and the CPU oracle's
statement of the same rule agree bit for bit, that every kernel variant agrees, and that with the extension
switched off the padding words of a material are ignored, as the shader ignores them."""
import numpy as np
import pytest

from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture()
def transmission(oracle):
    oracle.lib().orc_set_transmission(1)
    yield
    oracle.lib().orc_set_transmission(0)


def _glass_scene(rtmi):
    sc, pose = rtmi.scenes.stress_scene(5000, (50, 50), 4, seed=21, extent=4.0, glass=True)
    return sc, pose


def _render(rtmi, sc, pose, size, rpp, max_bounce, first, n, variant, accel, enable=True):
    fd = rtmi.make_frame_data(size[0], size[1], sc, pose=pose, rays_per_pixel=rpp, max_bounce=max_bounce)
    with rtmi.Renderer(size[0], size[1], variant=variant) as r:
        r.enable_transmission(enable)
        r.set_acceleration(accel)
        r.upload_scene(sc)
        r.enable_validation(True)
        r.render_frames(fd, first, n)
        return dict(accum=r.read_accum(), rgba8=r.read_output_rgba8(), prim_id=r.read_hit_ids(),
                    chain_hash=r.read_chain_hash(), rng_state=r.read_rng_state())


def _oracle(oracle, sc, pose, size, rpp, max_bounce, first, n):
    osc = oracle.Scene.from_product(sc)
    ofd = oracle.frame_data(size[0], size[1], pos=pose[0], yaw=pose[1], pitch=pose[2], rpp=rpp, max_bounce=max_bounce, scene=osc)
    return oracle.render(ofd, osc, first, n)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_ten_thousand_primitives_against_the_oracle(rtmi, oracle, transmission, variant):
    sc, pose = _glass_scene(rtmi)
    size = (48, 27)
    want = _oracle(oracle, sc, pose, size, 2, 9, 1, 3)
    got = _render(rtmi, sc, pose, size, 2, 9, 1, 3, variant, rtmi.abi.ACCEL_BVH)
    assert_same(got, want)


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_exhaustive_loops_and_every_variant(rtmi, oracle, transmission, variant):
    """A few hundred primitives without a hierarchy (variant 2, the compacting pair, has no transmitting material in
    its start records and must fall back by itself; AUTO must not start from stored primary hits)."""
    sc, pose = rtmi.scenes.stress_scene(150, (6, 6), 2, seed=4, extent=3.0, glass=True)
    size = (40, 24)
    want = _oracle(oracle, sc, pose, size, 1, 8, 1, 9)
    got = _render(rtmi, sc, pose, size, 1, 8, 1, 9, variant, rtmi.abi.ACCEL_NONE)
    assert_same(got, want)
    # the same launch again, continuing the sum over a second range (AUTO's second launch would be the one to load a store)
    fd = rtmi.make_frame_data(size[0], size[1], sc, pose=pose, rays_per_pixel=1, max_bounce=8)
    osc = oracle.Scene.from_product(sc)
    ofd = oracle.frame_data(size[0], size[1], pos=pose[0], yaw=pose[1], pitch=pose[2], rpp=1, max_bounce=8, scene=osc)
    want2 = oracle.render(ofd, osc, 10, 9, accum=want["accum"])
    with rtmi.Renderer(size[0], size[1], variant=variant) as r:
        r.enable_transmission(True)
        r.set_acceleration(rtmi.abi.ACCEL_NONE)
        r.upload_scene(sc)
        r.render_frames(fd, 1, 9)
        r.render_frames(fd, 10, 9)
        got2 = r.read_accum()
    assert np.array_equal(got2.view(np.uint32), want2["accum"].view(np.uint32))


def test_the_extension_changes_the_image_and_only_where_it_should(rtmi, oracle, transmission):
    sc, pose = _glass_scene(rtmi)
    plain, _ = rtmi.scenes.stress_scene(5000, (50, 50), 4, seed=21, extent=4.0, glass=False)
    size = (48, 27)
    on = _render(rtmi, sc, pose, size, 1, 9, 1, 2, 4, rtmi.abi.ACCEL_BVH)
    off = _render(rtmi, sc, pose, size, 1, 9, 1, 2, 4, rtmi.abi.ACCEL_BVH, enable=False)
    ref = _render(rtmi, plain, pose, size, 1, 9, 1, 2, 4, rtmi.abi.ACCEL_BVH)
    # switched off, the flags in the padding are ignored: the image of the plain scene, bit for bit
    assert_same(off, ref)
    # switched on, paths that meet a transmitting surface differ; primary hits do not
    assert not np.array_equal(on["accum"].view(np.uint32), ref["accum"].view(np.uint32))
    assert np.array_equal(on["prim_id"], ref["prim_id"])


def test_oracle_switch_is_off_by_default(rtmi, oracle):
    """Without orc_set_transmission the oracle ignores the padding too (what the shader does)."""
    sc, pose = rtmi.scenes.stress_scene(150, (6, 6), 2, seed=4, extent=3.0, glass=True)
    plain, _ = rtmi.scenes.stress_scene(150, (6, 6), 2, seed=4, extent=3.0, glass=False)
    a = _oracle(oracle, sc, pose, (32, 18), 1, 6, 1, 2)
    b = _oracle(oracle, plain, pose, (32, 18), 1, 6, 1, 2)
    assert np.array_equal(a["accum"].view(np.uint32), b["accum"].view(np.uint32))
