"""The hierarchy path WITHOUT a GPU: the library's own traversal code and scene packing, compiled for the
host (tests/emu), against the oracle's exhaustive loops (Raytracing.comp:338-388), ray by ray and bit for bit.
The same comparison runs on the device in tests/test_gpu_bvh*.py; here it also runs under AddressSanitizer."""
import os
import subprocess
import sys

import numpy as np
import pytest

import emu_lib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))


@pytest.fixture(scope="module")
def rtmi_host():
    # only host-side entry points of the package are used here (scene records); no device is touched
    import raytracer_amd
    return raytracer_amd


def _same(a, b):
    return np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("kind", ["strict", "fma"])
def test_small_stress_scene_against_the_oracle(rtmi_host, oracle, kind):
    sc, _ = rtmi_host.scenes.stress_scene(1500, (20, 20), 3, seed=11, extent=4.0)
    osc = oracle.Scene.from_product(sc)
    rs = np.random.RandomState(4)
    n = 4000
    o = rs.uniform(-5, 5, size=(n, 3)).astype(np.float32)
    d = (rs.normal(size=(n, 3)) * rs.uniform(0.3, 1.4, size=(n, 1))).astype(np.float32)
    want = oracle.closest_hit(osc, o, d, profile=kind)
    assert (want[1] != 0xFFFFFFFF).mean() > 0.3
    assert _same(emu_lib.closest_hit(sc, o, d, False, kind=kind), want)      # the library's exhaustive loops
    assert _same(emu_lib.closest_hit(sc, o, d, True, kind=kind), want)       # and its hierarchy


def _scenes_and_rays(n):
    import grazing_rate as G
    rs = np.random.RandomState(21)
    out = []
    small = G.big_triangles(rs, 500, (0.05, 0.4))
    mixed = np.concatenate([small[:250], G.big_triangles(rs, 20), small[250:]])   # a few large triangles among small ones
    for name, tri in (("noisy_grid", G.noisy_grid(rs, 40, 20)), ("big_triangles", G.big_triangles(rs, 150)),
                      ("mid_triangles", G.big_triangles(rs, 400, (0.3, 2.0))), ("small_triangles", G.big_triangles(rs, 800, (0.05, 0.4))),
                      ("mixed", mixed)):
        sc = G.one_mesh_scene(tri)
        rays = [gen(rs, tri, n) for gen in (G.rays_uniform, G.rays_far, G.rays_grazing)]
        out.append((name, sc, np.concatenate([r[0] for r in rays]), np.concatenate([r[1] for r in rays])))
    return out


def test_rays_along_triangle_planes_hierarchy_equals_exhaustive(rtmi_host, oracle):
    """Meshes of large, middle-sized and small triangles, and rays aimed to run almost inside the plane of a triangle
    and pass near it: the hierarchy must return what the exhaustive loops return (with the proven margins, cap 1)."""
    for name, sc, o, d in _scenes_and_rays(6000):
        osc = oracle.Scene.from_product(sc)
        want = oracle.closest_hit(osc, o, d, profile="strict")
        got = emu_lib.closest_hit(sc, o, d, True, cap=1.0)
        differ = (got[1] != want[1]) | (got[0].view(np.uint32) != want[0].view(np.uint32))
        assert differ.sum() == 0, "%s: %d of %d rays differ" % (name, int(differ.sum()), len(o))
        assert (want[1] != 0xFFFFFFFF).mean() > 0.02, name


def test_traversal_under_address_sanitizer(rtmi_host):
    """The same traversal in a process of its own with AddressSanitizer and UBSan: every read of the packed scene
    is checked against its extent (the emulation binds an exact-size copy)."""
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no AddressSanitizer runtime")
    emu_lib.build()
    code = (
        "import sys, numpy as np\n"
        "sys.path[:0] = [%r, %r, %r]\n"
        "import emu_lib, raytracer_amd as R\n"
        "import test_emu_bvh as T\n"
        "for name, sc, o, d in T._scenes_and_rays(1500):\n"
        "    a = emu_lib.closest_hit(sc, o, d, True, kind='asan')\n"
        "    b = emu_lib.closest_hit(sc, o, d, False, kind='asan')\n"
        "    print(name, int((a[1] != b[1]).sum()))\n"
        "sc, _ = R.scenes.stress_scene(800, (12, 12), 70, seed=2, extent=3.0)\n"   # more than 64 meshes: gate chunks
        "rs = np.random.RandomState(0)\n"
        "o = rs.uniform(-4, 4, size=(1500, 3)); d = rs.normal(size=(1500, 3))\n"
        "a = emu_lib.closest_hit(sc, o, d, True, kind='asan'); b = emu_lib.closest_hit(sc, o, d, False, kind='asan')\n"
        "print('stress', int((a[1] != b[1]).sum()))\n"
    ) % (HERE, os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "tools"))
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0", RTMI_HOST_ONLY="1",
               RTMI_LIBRARY=os.path.join(os.path.dirname(HERE), "raytracer_amd", "librtmi_host_asan.so"))
    if not os.path.exists(env["RTMI_LIBRARY"]):
        subprocess.check_call(["make", "-C", os.path.join(os.path.dirname(HERE), "raytracer_amd", "csrc"), "host-asan"],
                              stdout=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 6 and all(line.split()[1] == "0" for line in lines), out.stdout
