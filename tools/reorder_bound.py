"""What could re-ordering rays between bounces buy on the stress scene?  An upper bound, measured: the same set of
rays is walked by a kernel that does nothing but traverse (persistent waves, resumable walks, rays taken in the
order given: rtmi_selftest_walk) once in random order, as a path tracer's bounce rays arrive, and once sorted by
where they start and where they point (Morton code of the origin cell, then direction octant), which is the best
a wavefront design with a full sort between bounces could present to its traversal stage.  Two populations:
random rays inside the scene, and bounce-like rays (origins on the surfaces the first population hit, directions
random in the hemisphere around the ray that arrived).  The primary rays of the benchmark view, coherent by
construction, are timed as a third reference.

    python tools/reorder_bound.py [n_rays] [cell_bits]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 6


def spread(v, b):
    out = np.zeros(len(v), dtype=np.uint64)
    for k in range(b):
        out |= ((v >> np.uint64(k)) & np.uint64(1)) << np.uint64(3 * k)
    return out


def sort_key(o, d, e=50.0):
    q = np.clip(((o + e) / (2 * e) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1).astype(np.uint64)
    morton = (spread(q[:, 0], bits) << np.uint64(2)) | (spread(q[:, 1], bits) << np.uint64(1)) | spread(q[:, 2], bits)
    octant = ((d[:, 0] < 0).astype(np.uint64) << np.uint64(2)) | ((d[:, 1] < 0).astype(np.uint64) << np.uint64(1)) | (d[:, 2] < 0).astype(np.uint64)
    return (morton << np.uint64(3)) | octant


scene, pose = R.scenes.stress_scene()
rs = np.random.RandomState(1)
o = rs.uniform(-50, 50, size=(n, 3)).astype(np.float32)
d = rs.normal(size=(n, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
with R.Renderer(8, 8) as r:
    r.upload_scene(scene)
    nS, nM = scene.n_spheres, scene.n_meshes

    def timed(label, oo, dd):
        t, prim, ms = r.selftest_walk(nS, nM, oo, dd, repeats=4)
        print("%-46s %9d rays  %8.2f ms  %7.1f M rays/s  hit fraction %.3f" % (label, len(oo), ms, len(oo) / ms / 1e3,
                                                                             float((prim != R.abi.MISS_ID).mean())), flush=True)
        return t, prim, ms

    t, prim, ms_rand = timed("random rays, random order", o, d)
    order = np.argsort(sort_key(o, d), kind="stable")
    _, prim_s, ms_sort = timed("random rays, sorted by origin cell + octant", o[order], d[order])
    assert np.array_equal(prim_s, prim[order])
    print("  sorted / random: %.3f" % (ms_sort / ms_rand))
    # bounce-like rays: leave the surfaces that were hit, into the hemisphere the ray came from
    hit = prim != R.abi.MISS_ID
    o2 = (o[hit] + d[hit] * t[hit, None]).astype(np.float32)
    d2 = rs.normal(size=o2.shape).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    flip = np.einsum("ij,ij->i", d2, d[hit]) > 0
    d2[flip] = -d2[flip]
    _, prim2, ms_rand2 = timed("bounce-like rays, random order", o2, d2)
    order2 = np.argsort(sort_key(o2, d2), kind="stable")
    _, prim2_s, ms_sort2 = timed("bounce-like rays, sorted", o2[order2], d2[order2])
    assert np.array_equal(prim2_s, prim2[order2])
    print("  sorted / random: %.3f" % (ms_sort2 / ms_rand2))
    # the benchmark view's primary rays in tile order
    w, h = 1920, 1080
    fd = R.make_frame_data(w, h, scene, pose, 1, 16)
    inv_p = np.array(fd.camera_inverse_projection, dtype=np.float64).reshape(4, 4)
    inv_v = np.array(fd.camera_inverse_view, dtype=np.float64).reshape(4, 4)
    ys, xs = np.mgrid[0:h, 0:w]
    cx, cy = xs / w * 2 - 1, ys / h * 2 - 1
    tgt = np.stack([cx, cy, np.ones_like(cx), np.ones_like(cx)], axis=-1) @ inv_p     # column-major storage: row vector times it
    v = tgt[..., :3] / tgt[..., 3:4]
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    dirs = (np.concatenate([v, np.zeros_like(v[..., :1])], axis=-1) @ inv_v)[..., :3]
    tiles = dirs.reshape(h // 8, 8, w // 8, 8, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 3).astype(np.float32)
    op = np.tile(np.array(fd.camera_pos[:3], dtype=np.float32), (len(tiles), 1))
    timed("primary rays of the benchmark view, tile order", op, tiles)
