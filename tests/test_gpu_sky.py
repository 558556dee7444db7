"""GetEnvironmentLight (Raytracing.comp:390-406) on the GPU, both the per-lane routine and the one that
shares a wave's pow evaluations out over its lanes, against the oracle's, bit for bit: the boundaries
at which the wave routine stops evaluating terms whose value it knows (d.y = 0, 0.4, -0.01 and the
floats next to them, directions at right angles to the sun), special values, and sun exponents that
switch the shortcuts off."""
import ctypes as C

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _directions():
    rs = np.random.RandomState(77)
    f = np.float32
    edges = []
    for v in (0.0, 0.4, -0.01, 0.2, 1.0, -1.0, 1e-30, -1e-30, 0.39999, 0.40001, -0.00999, -0.01001):
        b = np.array([v], dtype=f)
        edges += [b[0], np.nextafter(b[0], f(np.inf)), np.nextafter(b[0], f(-np.inf))]
    edges += [f(-0.0), f(np.inf), f(-np.inf), f(np.nan), f(1e-45), f(-1e-45), f(3e38)]
    ys = np.array(edges, dtype=f)
    d_edges = np.stack([np.tile(rs.uniform(-1, 1, 3).astype(f), len(ys) // 3 + 1)[:len(ys)], ys,
                        rs.uniform(-1, 1, len(ys)).astype(f)], axis=1)
    unit = rs.normal(size=(6000, 3)).astype(f)
    unit /= np.linalg.norm(unit, axis=1, keepdims=True).astype(f)
    short = unit * rs.uniform(0.05, 1.0, size=(len(unit), 1)).astype(f)        # mixes of unit vectors are shorter than 1
    flat = unit.copy()
    flat[:, 1] = rs.uniform(-0.02, 0.01, len(flat)).astype(f)                   # around the ground / sky transition
    # at right angles to the sun (dot = 0 up to rounding) and exactly away from it
    sun = np.array([-0.4, -0.4, -0.4], dtype=f)                                 # main.cpp:123; -(sun.xzy) is what the shader dots with
    neg = -sun[[0, 2, 1]]
    perp = np.cross(unit[:500], neg).astype(f)
    special = np.array([[np.nan, 0.5, 0.1], [0.1, 0.5, np.inf], [0, 0, 0], [-0.0, -0.0, -0.0], [1e20, 1e20, 1e20],
                        [-1e20, 0.3, 1e20], [np.inf, np.inf, np.inf], [1e-40, 1e-40, 1e-40]], dtype=f)
    return np.concatenate([d_edges, unit, short, flat, perp, -neg[None, :], neg[None, :], special]).astype(f)


@pytest.mark.parametrize("focus,intensity", [(None, None), (1.0, 2.0), (0.0, 1.0), (-2.0, 1.0), (300.0, 1e30), (0.5, -3.0)])
def test_sky_routines_match_the_oracle(rtmi, oracle, focus, intensity):
    sc = rtmi.sphere_scene()
    fd = rtmi.make_frame_data(64, 64, sc, rtmi.BENCH_POSE, 1, 4)
    ofd = oracle.frame_data(64, 64)
    if focus is not None:
        fd.sun_focus, fd.sun_intensity = focus, intensity
        ofd.sun_focus, ofd.sun_intensity = focus, intensity
    assert bytes(fd)[176:248] == bytes(ofd)[176:248]          # same sky constants on both sides
    dirs = _directions()
    want = np.empty_like(dirs)
    rgb = (C.c_float * 3)()
    L = oracle.lib()
    for i, d in enumerate(dirs):
        L.orc_env_light(C.byref(ofd), oracle.fl3(*[float(v) for v in d]), rgb)
        want[i] = rgb[:]
    with rtmi.Renderer(64, 64) as r:
        plain = r.selftest_env_light(fd, dirs, False)
        wave = r.selftest_env_light(fd, dirs, True)
    bad = [i for i in range(len(dirs)) if not cases.bits_equal(plain[i], want[i])]
    assert not bad, "per-lane routine differs from the oracle for %d directions, first %r" % (len(bad), dirs[bad[0]])
    bad = [i for i in range(len(dirs)) if not cases.bits_equal(wave[i], want[i])]
    assert not bad, "wave routine differs from the oracle for %d directions, first %r: %r vs %r" % (
        len(bad), dirs[bad[0]], wave[bad[0]], want[bad[0]])


def test_sky_with_more_than_64_pow_requests_per_wave(rtmi, oracle):
    """Directions that all need both pow evaluations (0 < d.y < 0.4, facing the sun): 42 lanes x 2 requests."""
    sc = rtmi.sphere_scene()
    fd = rtmi.make_frame_data(64, 64, sc, rtmi.BENCH_POSE, 1, 4)
    ofd = oracle.frame_data(64, 64)
    rs = np.random.RandomState(3)
    n = 43 * 50
    dirs = np.stack([rs.uniform(0.1, 0.6, n), rs.uniform(0.01, 0.39, n), rs.uniform(0.2, 0.9, n)], axis=1).astype(np.float32)
    want = np.empty_like(dirs)
    rgb = (C.c_float * 3)()
    for i, d in enumerate(dirs):
        oracle.lib().orc_env_light(C.byref(ofd), oracle.fl3(*[float(v) for v in d]), rgb)
        want[i] = rgb[:]
    with rtmi.Renderer(64, 64) as r:
        wave = r.selftest_env_light(fd, dirs, True)
    assert cases.bits_equal(wave, want)
