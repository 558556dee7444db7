"""How often the hierarchy and the exhaustive loops disagree on a ray, by population of rays.

ClosestHit of the same rays is evaluated twice on the GPU (rtmi_selftest_closest_hit): once with
RTMI_ACCEL_NONE (the shader's loops, Raytracing.comp:338-388) and once through the hierarchy.  Scenes:
a displaced grid with the local geometry of the stress scene's mesh (cells 0.1 x 0.2, white-noise
heights), and a set of large triangles in random orientation.  Ray populations: uniform, far origins,
and rays aimed to run almost inside the plane of a triangle and pass near it.

    python tools/grazing_rate.py [--rays 1000000] [--out file]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rtmi  # noqa: E402

F = np.float32


def noisy_grid(rs, nx=100, nz=50, cell=(0.1, 0.2), noise=0.2, amp=2.0):
    xs = (np.arange(nx + 1) - nx / 2) * cell[0]
    zs = (np.arange(nz + 1) - nz / 2) * cell[1]
    X, Z = np.meshgrid(xs, zs)
    Y = amp * np.sin(6 * np.pi * X / 100.0) * np.cos(4 * np.pi * Z / 100.0) + noise * rs.uniform(-1, 1, size=X.shape)
    P = np.stack([X, Y, Z], axis=-1).astype(F)
    p00, p10, p01, p11 = P[:-1, :-1], P[:-1, 1:], P[1:, :-1], P[1:, 1:]
    tri = np.zeros((nz, nx, 2), dtype=rtmi.host.TRIANGLE_DTYPE)
    for name, a, b in (("p1", p00, p00), ("p2", p01, p11), ("p3", p11, p10)):
        tri[name][:, :, 0, :3] = a
        tri[name][:, :, 1, :3] = b
    tri = tri.reshape(-1)
    for name in ("n1", "n2", "n3"):
        tri[name][:, 1] = 1.0
    return tri


def big_triangles(rs, n_tri=400, sizes=(2.0, 10.0)):
    centre = rs.uniform(-20, 20, size=(n_tri, 3))
    e1 = rs.normal(size=(n_tri, 3))
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = np.cross(e1, rs.normal(size=(n_tri, 3)))
    e2 /= np.linalg.norm(e2, axis=1, keepdims=True)
    size = rs.uniform(sizes[0], sizes[1], size=(n_tri, 1))
    tri = np.zeros(n_tri, dtype=rtmi.host.TRIANGLE_DTYPE)
    tri["p1"][:, :3] = (centre - 0.5 * size * e1 - 0.3 * size * e2).astype(F)
    tri["p2"][:, :3] = (centre + 0.5 * size * e1 - 0.3 * size * e2).astype(F)
    tri["p3"][:, :3] = (centre + 0.6 * size * e2).astype(F)
    for name in ("n1", "n2", "n3"):
        tri[name][:, 1] = 1.0
    return tri


def one_mesh_scene(tri):
    v = np.concatenate([tri["p1"][:, :3], tri["p2"][:, :3], tri["p3"][:, :3]]).astype(np.float64)
    mesh = np.zeros(1, dtype=rtmi.host.MESH_DTYPE)
    mesh["start_triangle"], mesh["num_triangles"] = 0, len(tri)
    mesh["bounding_point1"][0, :3] = (v.min(axis=0) - 1.0).astype(F)
    mesh["bounding_point2"][0, :3] = (v.max(axis=0) + 1.0).astype(F)
    mesh["material"]["color"][0] = (1.0, 1.0, 1.0)
    return rtmi.Scene.from_records(np.zeros(0, dtype=rtmi.host.SPHERE_DTYPE), tri, mesh)


def unit(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def rays_uniform(rs, tri, n):
    v = tri["p1"][:, :3].astype(np.float64)
    lo, hi = v.min(axis=0) - 3.0, v.max(axis=0) + 3.0
    o = rs.uniform(lo, hi, size=(n, 3))
    d = unit(rs.normal(size=(n, 3))) * rs.uniform(0.3, 1.4, size=(n, 1))
    return o, d


def rays_far(rs, tri, n):
    k = rs.randint(0, len(tri), n)
    target = tri["p1"][k, :3].astype(np.float64) + rs.normal(size=(n, 3)) * 0.05
    d = unit(rs.normal(size=(n, 3)))
    d[:, 1] *= 0.2                                  # mostly shallow
    d = unit(d) * rs.uniform(0.3, 1.4, size=(n, 1))
    o = target - unit(d) * rs.uniform(5.0, 40.0, size=(n, 1))
    return o, d


def rays_grazing(rs, tri, n):
    """Almost inside the plane of triangle k, towards a point within a few edge lengths of it."""
    k = rs.randint(0, len(tri), n)
    p1, p2, p3 = (tri[name][k, :3].astype(np.float64) for name in ("p1", "p2", "p3"))
    e1, e2 = p2 - p1, p3 - p1
    nrm = unit(np.cross(e1, e2))
    size = np.maximum(np.linalg.norm(e1, axis=1), np.linalg.norm(e2, axis=1))[:, None]
    bu, bv = rs.uniform(-1.5, 2.5, size=(n, 1)), rs.uniform(-1.5, 2.5, size=(n, 1))
    target = p1 + bu * e1 + bv * e2
    along = unit(rs.normal(size=(n, 1)) * unit(e1) + rs.normal(size=(n, 1)) * unit(np.cross(nrm, e1)))
    tilt = 10.0 ** rs.uniform(-7, -2, size=(n, 1))
    d = along - tilt * nrm
    o = target - d * rs.uniform(1.0, 40.0, size=(n, 1)) * np.maximum(size, 0.3)
    return o, d * rs.choice([0.5, 1.0, 1.4], size=(n, 1))


CAP = None


def compare(scene, o, d, chunk=1 << 20):
    o32, d32 = o.astype(F), d.astype(F)
    res = {}
    for mode in (rtmi.abi.ACCEL_NONE, rtmi.abi.ACCEL_BVH):
        with rtmi.Renderer(8, 8) as r:
            r.set_acceleration(mode)
            if CAP is not None:
                r.set_grazing_cap(CAP)
            r.upload_scene(scene)
            ts, qs = [], []
            for s in range(0, len(o32), chunk):
                t, q = r.selftest_closest_hit(0, 1, o32[s:s + chunk], d32[s:s + chunk])
                ts.append(t)
                qs.append(q)
            res[mode] = (np.concatenate(ts), np.concatenate(qs))
    (t0, q0), (t1, q1) = res[rtmi.abi.ACCEL_NONE], res[rtmi.abi.ACCEL_BVH]
    differ = (q0 != q1) | (t0.view(np.uint32) != t1.view(np.uint32))
    return differ, q0, q1, t0, t1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1_000_000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--cap", type=float, default=None, help="rtmi_set_grazing_cap value (default: the library's)")
    ap.add_argument("--scenes", default="", help="comma-separated subset of the scenes")
    a = ap.parse_args()
    global CAP
    CAP = a.cap
    rs = np.random.RandomState(99)
    report = {}
    scenes = (("noisy_grid", noisy_grid(rs)), ("big_triangles", big_triangles(rs)),
              ("mid_triangles", big_triangles(rs, 2000, (0.3, 2.0))), ("small_triangles", big_triangles(rs, 8000, (0.05, 0.4))))
    for scene_name, tri in scenes:
        if a.scenes and scene_name not in a.scenes.split(","):
            continue
        print("scene", scene_name, len(tri), "triangles", flush=True)
        sc = one_mesh_scene(tri)
        for pop_name, gen in (("uniform", rays_uniform), ("far", rays_far), ("grazing", rays_grazing)):
            o, d = gen(rs, tri, a.rays)
            differ, q0, q1, t0, t1 = compare(sc, o, d)
            hits = int((q0 != rtmi.abi.MISS_ID).sum())
            lost = int((differ & (q0 != rtmi.abi.MISS_ID) & ((q1 == rtmi.abi.MISS_ID) | (t1 > t0))).sum())
            entry = dict(rays=int(len(o)), hits_exhaustive=hits, differ=int(differ.sum()), hierarchy_missed_a_hit=lost)
            report["%s/%s" % (scene_name, pop_name)] = entry
            print(scene_name, pop_name, entry, flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
