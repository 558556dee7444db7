"""Helper of test_gpu_reduce_stub.py, run as a process of its own with RTMI_RCCL_LIBRARY naming the tests'
stand-in for RCCL: the in-process multi-context sum (rtmi_reduce_accum) with all contexts on GPU 0, compared
bit for bit with the float sum, in rank order, of the partial images.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rtmi  # noqa: E402

W, H, FRAMES = 96, 54, 11


def partial_ranges(world):
    q, rem = divmod(FRAMES, world)
    return [(1 + g * q + min(g, rem), q + (1 if g < rem else 0)) for g in range(world)]


def run(world, root, scene, fd, lib):
    rs = [rtmi.Renderer(W, H) for _ in range(world)]
    try:
        for r in rs:
            r.upload_scene(scene)
            r.write_accum(np.full((H, W, 4), 7.0, np.float32))   # junk that the clear must remove
        parts = []
        for r, (first, n) in zip(rs, partial_ranges(world)):
            r.clear_accum()
            r.render_frames(fd, first, n)
        # the partial images, from contexts that take no part in the sum
        for first, n in partial_ranges(world):
            with rtmi.Renderer(W, H) as plain:
                plain.upload_scene(scene)
                plain.clear_accum()
                plain.render_frames(fd, first, n)
                parts.append(plain.read_accum())
        arr = (C.c_void_p * world)(*[r._ctx for r in rs])
        rc = lib.rtmi_reduce_accum(C.cast(arr, C.POINTER(C.c_void_p)), world, root)
        assert rc == 0, (rc, lib.rtmi_last_error(rs[root]._ctx))
        rs[root].resolve_output(FRAMES)
        for r in rs:
            r.synchronize()
        want = parts[0].copy()
        for p in parts[1:]:
            want = want + p
        got = rs[root].read_accum()
        same_sum = bool(np.array_equal(got[..., :3].view(np.uint32), want[..., :3].view(np.uint32)))
        alpha_one = bool((got[..., 3] == 1.0).all())
        # a rank that is not the root keeps its partial image
        other = (root + 1) % world
        kept = bool(np.array_equal(rs[other].read_accum().view(np.uint32), parts[other].view(np.uint32)))
        return dict(world=world, root=root, same_sum=same_sum, alpha_one=alpha_one, other_rank_kept=kept)
    finally:
        for r in rs:
            r.close()


def main():
    lib = rtmi.abi.load()
    scene = rtmi.reference_scene()
    fd = rtmi.make_frame_data(W, H, scene, rtmi.BENCH_POSE, rays_per_pixel=2, max_bounce=5)
    out = dict(stub=int(lib.rtmi_reduce_same_device_allowed()), runs=[])
    out["runs"].append(run(3, 1, scene, fd, lib))
    out["runs"].append(run(2, 0, scene, fd, lib))      # another device list: the kept communicators are replaced
    out["shutdown"] = int(lib.rtmi_shutdown())
    out["runs"].append(run(2, 1, scene, fd, lib))      # and the library opens again after a shutdown
    print(json.dumps(out))


if __name__ == "__main__":
    main()
