"""GPU tests of the fused arithmetic profile (rtmi_set_profile(RTMI_PROFILE_FMA); run on an MI355X with
`-m gpu`).

The profile has its own CPU oracle build (oracle/liboracle_fma.so, the same file with fmaf at the same
fixed list of sites) and its own fixtures (tests/golden/*_fma.npz); within the profile the bar is the
strict profile's: every output bit for bit.  Across profiles the results differ by rounding only, which
the last test measures.  Running the whole GPU suite with RTMI_PROFILE=fma in the environment makes every
other test use this profile on both sides as well."""
import numpy as np
import pytest

import cases
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu

VARIANTS = [0, 1, 2, 3, 4]


def _render(rtmi, case, variant, profile, validate=True, accel=None):
    sc = cases.product_scene(case)
    fd = cases.product_frame_data(case, sc)
    w, h = case["size"]
    with rtmi.Renderer(w, h, variant=variant, profile=profile) as r:
        if accel is not None:
            r.set_acceleration(accel)
        r.upload_scene(sc)
        assert r.profile() == profile
        r.enable_validation(validate)
        if case.get("seed_accum"):
            r.write_accum(cases.seed_accum(case))
        r.render_frames(fd, case["first"], case["n"])
        out = dict(accum=r.read_accum(), rgba8=r.read_output_rgba8())
        if validate:
            out.update(prim_id=r.read_hit_ids(), chain_hash=r.read_chain_hash(), rng_state=r.read_rng_state(),
                       counters=r.counters())
    return out


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_fused_profile_against_its_fixtures_and_oracle(rtmi, oracle, name, variant):
    case = cases.CASES[name]
    got = _render(rtmi, case, variant, "fma")
    assert_same(got, cases.load_golden(name, "fma"))
    assert_same(got, cases.run_oracle(case, profile="fma"))


@pytest.mark.parametrize("variant", [1, 2])
def test_both_profiles_in_one_process(rtmi, variant):
    """Contexts of both profiles side by side: each reproduces its own fixtures, also through the
    hierarchy and through the uninstrumented kernels; the images are not the same bits."""
    name = "c3_mesh_bench_pose"
    case = cases.CASES[name]
    sc = cases.product_scene(case)
    fd = cases.product_frame_data(case, sc)
    w, h = case["size"]
    with rtmi.Renderer(w, h, variant=variant, profile="strict") as a, rtmi.Renderer(w, h, variant=variant, profile="fma") as b:
        b.set_acceleration(rtmi.abi.ACCEL_BVH)
        for r in (a, b):
            r.upload_scene(sc)
        for r in (b, a, b):
            r.render_frames(fd, case["first"], case["n"])
        ia, ib = a.read_accum(), b.read_accum()
        assert a.profile() == "strict" and b.profile() == "fma"
        # the profile of an uploaded scene stays until the next upload
        b.set_profile("strict")
        assert b.profile() == "fma"
        b.upload_scene(sc)
        assert b.profile() == "strict"
        b.render_frames(fd, case["first"], case["n"])
        ic = b.read_accum()
        with pytest.raises(rtmi.RtmiError):
            b.set_profile(5)
    assert cases.bits_equal(ia, cases.load_golden(name, "strict")["accum"])
    assert cases.bits_equal(ib, cases.load_golden(name, "fma")["accum"])
    assert cases.bits_equal(ic, ia) and not cases.bits_equal(ia, ib)


def test_fused_profile_math_hooks(rtmi, oracle):
    """The profile's pow polynomials and sky term on arbitrary arguments (per-lane and wave-wide routines)."""
    import ctypes as C
    L = oracle.lib("fma")
    rs = np.random.RandomState(11)
    x = np.concatenate([rs.uniform(0.0, 1.0, 4000), 10.0 ** rs.uniform(-40, 38, 2000), [0.0, 1.0, 0.5, 2.0, np.inf, np.nan, -1.0, 1e-45]]).astype(np.float32)
    y = np.concatenate([rs.uniform(-3.0, 60.0, len(x) - 4), [0.35, 50.0, 0.0, -0.0]]).astype(np.float32)
    e = np.concatenate([rs.uniform(-160.0, 140.0, 5000), [0.0, -0.0, 127.5, -149.5, 128.5, np.nan]]).astype(np.float32)
    with rtmi.Renderer(8, 8, profile="fma") as r:
        got_log, got_exp, got_pow = r.selftest_math(2, x), r.selftest_math(3, e), r.selftest_math(5, x, y)
        fd = rtmi.make_frame_data(64, 36, rtmi.sphere_scene(), cases.BENCH_POSE)
        d = rs.normal(size=(6000, 3)).astype(np.float32)
        d[::7, 1] = rs.uniform(-0.012, 0.002, size=len(d[::7])).astype(np.float32)     # around the ground / sky blend
        d[::11, 1] = rs.uniform(0.0, 0.45, size=len(d[::11])).astype(np.float32)       # the gradient's range
        sky_lane, sky_wave = r.selftest_env_light(fd, d, False), r.selftest_env_light(fd, d, True)
    want_log = np.array([L.orc_log2(float(v)) for v in x], dtype=np.float32)
    want_exp = np.array([L.orc_exp2(float(v)) for v in e], dtype=np.float32)
    want_pow = np.array([L.orc_pow(float(a), float(b)) for a, b in zip(x, y)], dtype=np.float32)
    assert cases.bits_equal(got_log, want_log) and cases.bits_equal(got_exp, want_exp) and cases.bits_equal(got_pow, want_pow)
    ofd = oracle.FrameData.from_buffer_copy(bytes(fd))
    want_sky = np.empty_like(d)
    out = (C.c_float * 3)()
    for i in range(len(d)):
        L.orc_env_light(C.byref(ofd), oracle.fl3(*[float(v) for v in d[i]]), out)
        want_sky[i] = out[:]
    assert cases.bits_equal(sky_lane, want_sky) and cases.bits_equal(sky_wave, want_sky)
    # and the strict library's polynomials give other bits for some of these arguments
    S = oracle.lib("strict")
    assert any(np.float32(S.orc_log2(float(v))).view(np.uint32) != w.view(np.uint32) for v, w in zip(x[:4000], want_log[:4000]))


def test_fused_profile_through_the_hierarchy(rtmi, oracle):
    """Closest hits and a rendered image of a 10^4-primitive scene: hierarchy kernels of the fused profile
    against the fused oracle's exhaustive loops."""
    sc, pose = rtmi.scenes.stress_scene(n_spheres=4000, grid=(40, 40), n_meshes=4, seed=777, extent=5.0)
    osc = oracle.Scene.from_product(sc)
    rs = np.random.RandomState(5)
    n = 20000
    o = rs.uniform(-6.5, 6.5, size=(n, 3)).astype(np.float32)
    d = (rs.uniform(-5.0, 5.0, size=(n, 3)) - o).astype(np.float32) * rs.uniform(0.01, 2.0, size=(n, 1)).astype(np.float32)
    d[::13, 0] = 0.0
    with rtmi.Renderer(8, 8, profile="fma") as r:
        r.set_acceleration(rtmi.abi.ACCEL_BVH)
        r.upload_scene(sc)
        t, prim = r.selftest_closest_hit(sc.n_spheres, sc.n_meshes, o, d)
    wt, wprim = oracle.closest_hit(osc, o, d, profile="fma")
    assert np.array_equal(prim, wprim) and cases.bits_equal(t, wt)
    assert int((prim != rtmi.abi.MISS_ID).sum()) > 3000

    w, h = 96, 54
    fd = rtmi.make_frame_data(w, h, sc, pose, rays_per_pixel=1, max_bounce=8)
    ofd = oracle.FrameData.from_buffer_copy(bytes(fd))
    want = oracle.render(ofd, osc, 1, 2, profile="fma")
    for variant in (2, 4):
        with rtmi.Renderer(w, h, variant=variant, profile="fma") as r:
            r.set_acceleration(rtmi.abi.ACCEL_BVH)
            r.upload_scene(sc)
            r.enable_validation(True)
            r.render_frames(fd, 1, 2)
            got = dict(accum=r.read_accum(), rgba8=r.read_output_rgba8(), prim_id=r.read_hit_ids(),
                       chain_hash=r.read_chain_hash(), rng_state=r.read_rng_state())
        assert_same(got, want)


def test_profiles_differ_by_rounding_only(rtmi):
    """Strict against fused on the reference scene at 1920 x 1080, 64 frames: primary hits agree almost
    everywhere, few paths take another course, and the mean radiance images agree to well over 50 dB."""
    w, h, frames = 1920, 1080, 64
    sc = rtmi.reference_scene()
    fd = rtmi.make_frame_data(w, h, sc, cases.BENCH_POSE, rays_per_pixel=1, max_bounce=8)
    out = {}
    for profile in ("strict", "fma"):
        with rtmi.Renderer(w, h, variant=2, profile=profile) as r:
            r.upload_scene(sc)
            r.enable_validation(True)
            r.render_frames(fd, 1, frames)
            out[profile] = (r.read_accum(), r.read_hit_ids(), r.read_chain_hash())
    (a, pa, ca), (b, pb, cb) = out["strict"], out["fma"]
    n = w * h
    assert int((pa != pb).sum()) <= n // 10000                     # primary hit ids
    assert int((ca != cb).sum()) <= n // 50                        # pixels in which some path of the 64 differs
    ma = np.clip(a[..., :3] / np.float32(frames), 0.0, 1.0).astype(np.float64)
    mb = np.clip(b[..., :3] / np.float32(frames), 0.0, 1.0).astype(np.float64)
    mse = float(np.mean((ma - mb) ** 2))
    assert mse > 0.0 and 10.0 * np.log10(1.0 / mse) >= 50.0
