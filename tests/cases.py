"""Shared parity cases: each builds the same scene + frame data for the oracle and for the product
(through the C ABI host functions) and checks that the two host sides agree byte for byte."""
import ctypes as C
import os

import numpy as np

import oracle_lib as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

DEFAULT_POSE = ((0.0, 1.0, -1.0), -90.0, 0.0)
BENCH_POSE = ((0.0, 1.5, -4.5), 90.0, -15.0)

# name -> dict(scene, quad_rule, pose, size, rpp, max_bounce, first, n, seed_accum)
CASES = {
    "c1_plumbing_default_pose": dict(scene="reference", pose=DEFAULT_POSE, size=(64, 36), rpp=1, max_bounce=4, first=1, n=4),
    "c3_mesh_bench_pose": dict(scene="reference", pose=BENCH_POSE, size=(96, 54), rpp=1, max_bounce=8, first=1, n=8),
    "c2_spheres_bench_pose": dict(scene="spheres", pose=BENCH_POSE, size=(96, 54), rpp=1, max_bounce=8, first=1, n=8),
    "reference_defaults_rpp4_continue": dict(scene="reference", pose=BENCH_POSE, size=(64, 36), rpp=4, max_bounce=6, first=3, n=3,
                                             seed_accum=True),
    "mesh_fan_triangulation": dict(scene="reference", quad_rule=1, pose=BENCH_POSE, size=(64, 36), rpp=1, max_bounce=8, first=1, n=2),
    "ragged_size_rpp3": dict(scene="reference", pose=BENCH_POSE, size=(37, 19), rpp=3, max_bounce=5, first=1, n=2),
}


def sphere_table():
    from raytracer_amd import host
    return host._SPHERE_SCENE_TABLE


def oracle_scene(case):
    if case["scene"] == "reference":
        return O.reference_scene(case.get("quad_rule", 0))
    return O.Scene([O.make_sphere(c, r, O.material(col, light, sm)) for c, r, col, light, sm in sphere_table()])


def product_scene(case):
    import raytracer_amd as R
    if case["scene"] == "reference":
        return R.reference_scene(quad_rule=case.get("quad_rule", 0))
    return R.sphere_scene()


def seed_accum(case):
    """Deterministic non-trivial starting contents for cases that continue an accumulation."""
    w, h = case["size"]
    if not case.get("seed_accum"):
        return np.zeros((h, w, 4), dtype=np.float32)
    rs = np.random.RandomState(1234)
    a = rs.uniform(0.0, 2.0, size=(h, w, 4)).astype(np.float32)
    a[..., 3] = 1.0
    return a


def oracle_frame_data(case, scene):
    w, h = case["size"]
    pos, yaw, pitch = case["pose"]
    return O.frame_data(w, h, pos, yaw, pitch, rpp=case["rpp"], max_bounce=case["max_bounce"], scene=scene)


def product_frame_data(case, scene):
    import raytracer_amd as R
    w, h = case["size"]
    return R.make_frame_data(w, h, scene, case["pose"], case["rpp"], case["max_bounce"])


def run_oracle(case, n_threads=0, profile=None):
    sc = oracle_scene(case)
    fd = oracle_frame_data(case, sc)
    return O.render(fd, sc, case["first"], case["n"], accum=seed_accum(case), n_threads=n_threads, profile=profile)


def golden_path(name, profile=None):
    """Fixture of an arithmetic profile ("strict", "fma"; None = the default of oracle_lib)."""
    profile = profile or O.DEFAULT_PROFILE
    return os.path.join(GOLDEN_DIR, name + ("_fma" if profile == "fma" else "") + ".npz")


def load_golden(name, profile=None):
    with np.load(golden_path(name, profile), allow_pickle=False) as z:
        out = {k: z[k] for k in z.files}
    out["counters"] = dict(zip(O.COUNTER_NAMES, (int(v) for v in out.pop("counter_values"))))
    return out


def bits_equal(a, b):
    """Bit-for-bit equality of float32 arrays, except that any NaN matches any NaN."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    return np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])
