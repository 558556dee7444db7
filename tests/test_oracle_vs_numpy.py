"""Cross-check of the C oracle against a second restatement of the shader (tests/np_shader.py: numpy
float32, vectorised over pixels, written separately from the shader text).  Both follow the same
arithmetic contract, so accumulation images, first-hit ids and RNG states must agree bit for bit.
This does not pin the oracle to the reference (nothing here can run the shader); it guards against an
error of transcription in either restatement."""
import ctypes as C

import numpy as np
import pytest

import cases
import np_shader


def _records(osc, rtmi_host):
    ns, nt, nm = osc.n_spheres, osc.n_tris.value, osc.n_meshes.value
    sph = np.frombuffer(osc.spheres, dtype=rtmi_host.SPHERE_DTYPE, count=ns)
    tri = np.frombuffer(osc.tris, dtype=rtmi_host.TRIANGLE_DTYPE, count=nt)
    msh = np.frombuffer(osc.meshes, dtype=rtmi_host.MESH_DTYPE, count=nm)
    return sph, tri, msh


def _pow_fn(oracle, profile=None):
    lib = oracle.lib(profile)

    def f(x, y):
        x = np.asarray(x, dtype=np.float32)
        y = np.broadcast_to(np.asarray(y, dtype=np.float32), x.shape)
        out = np.empty(x.shape, dtype=np.float32)
        for i in range(x.size):
            out.flat[i] = lib.orc_pow(float(x.flat[i]), float(y.flat[i]))
        return out
    return f


@pytest.mark.parametrize("scene,rpp,depth,first,n", [
    ("reference", 2, 5, 1, 2),      # sphere + two meshes, reset frame then an accumulating one
    ("spheres", 1, 8, 3, 2),        # spheres only, continuing a sum
    ("reference", 3, 1, 7, 1),      # single-segment paths, odd sample count
])
@pytest.mark.parametrize("profile", ["strict", "fma"])
def test_oracle_matches_numpy_restatement(oracle, rtmi, scene, rpp, depth, first, n, profile):
    w, h = 24, 14
    case = dict(scene=scene, pose=cases.BENCH_POSE, size=(w, h), rpp=rpp, max_bounce=depth)
    osc = cases.oracle_scene(case)
    fd = cases.oracle_frame_data(case, osc)
    rs = np.random.RandomState(4)
    seed = rs.uniform(0.0, 2.0, size=(h, w, 4)).astype(np.float32)
    seed[..., 3] = 1.0
    want = oracle.render(fd, osc, first, n, accum=seed, profile=profile)

    sph, tri, msh = _records(osc, rtmi.host)
    sh = np_shader.Shader(fd, sph, tri, msh, _pow_fn(oracle, profile), fused=profile == "fma")
    accum = seed.copy()
    with np.errstate(all="ignore"):
        for f in range(first, first + n):
            accum, prim, rng = sh.frame(accum, f)
    same = cases.bits_equal(accum, want["accum"])
    assert same, "accumulation differs in %d pixels" % int((accum != want["accum"]).any(axis=-1).sum())
    assert np.array_equal(prim, want["prim_id"])
    assert np.array_equal(rng, want["rng_state"])
    assert int((prim != oracle.MISS_ID).sum()) > w * h // 8      # the view does contain geometry
