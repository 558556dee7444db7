"""One rank of a multi-rank check of raytracer_amd.distributed.ShardedRenderer over the HIP library.

Launched once per rank (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment) by
tests/test_gpu_multi.py.  --backend nccl puts rank r on GPU r (RCCL); --backend gloo puts every rank
on GPU 0 (a one-GPU box: RCCL refuses two ranks on one device), which exercises the same sharding,
stream ordering and resolve code.

Each rank runs three steps of render + sum (+ resolve on the ranks that hold the sum) back to back,
with nothing between them that waits for the device, then checks:
  * its partial image, rendered on its own, equals what a plain context renders for its frame range,
    bit for bit;
  * the summed image equals the float32 sum of the ranks' partial images taken in rank order, bit
    for bit for two ranks (a two-term float sum does not depend on its order), within a few ulp per
    rank otherwise, and is within the documented tolerance of a one-context render of all frames;
  * alpha is back to 1 and the display image is sum / total_frames.
Prints "OK rank r" or "FAIL ..." and exits 0 / 1."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch
import torch.distributed as dist

import raytracer_amd as R
from raytracer_amd.distributed import ShardedRenderer, split_frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--dst", type=int, default=-1, help="-1: all-reduce; otherwise reduce onto this rank")
    ap.add_argument("--size", default="320x180")
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--scene", default="spheres", choices=["spheres", "reference"])
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    gpu = rank if args.backend == "nccl" else 0
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    if args.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h = (int(v) for v in args.size.split("x"))
    total = args.frames
    dst = None if args.dst < 0 else args.dst
    scene = R.sphere_scene() if args.scene == "spheres" else R.reference_scene()
    fd = R.make_frame_data(w, h, scene, R.BENCH_POSE, 1, 8)
    problems = []
    with R.Renderer(w, h, device=gpu, variant=2) as r:
        r.upload_scene(scene)
        sr = ShardedRenderer(r, dev)
        for _ in range(3):
            sr.accum.fill_(3.0)          # junk on the caller's stream: render() must order itself after it
            first, n = sr.render(fd, total, dst=dst)
            if dst is None or rank == dst:
                sr.resolve(total)
        holds_sum = dst is None or rank == dst
        summed = sr.accum.cpu().numpy()   # on the caller's stream, which render()/resolve() joined
        rgba8 = r.read_output_rgba8()
        sr.close()
    assert (first, n) == split_frames(total, world, rank)
    # every rank's partial image from a plain context (internal buffer, own stream)
    parts = []
    with R.Renderer(w, h, device=gpu, variant=2) as plain:
        plain.upload_scene(scene)
        for q in range(world):
            f0, nq = split_frames(total, world, q)
            plain.clear_accum()
            if nq:
                plain.render_frames(fd, f0, nq)   # adds into the cleared image; frame 1 overwrites
            a = plain.read_accum()
            if nq == 0:
                a[...] = 0.0
            parts.append(a)
        plain.render_frames(fd, 1, total)
        whole = plain.read_accum()
    if holds_sum:
        want = parts[0][..., :3].copy()
        for q in range(1, world):
            want = want + parts[q][..., :3]
        got = summed[..., :3]
        if world == 2:
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                problems.append("sum of two partial images differs in %d values" % int((got != want).sum()))
        elif not np.allclose(got, want, rtol=0, atol=world * 2e-7 * max(1.0, float(want.max()))):
            problems.append("sum differs from the rank-ordered sum: max abs %g" % float(np.abs(got - want).max()))
        tol = total * 4e-7 * max(1.0, float(whole.max()))
        if not np.allclose(got, whole[..., :3], rtol=0, atol=tol):
            problems.append("sum differs from the one-context render: max abs %g > %g" % (float(np.abs(got - whole[..., :3]).max()), tol))
        # the stated tolerance at equal samples per pixel: maximum absolute error of the mean radiance and PSNR
        # (peak 1, values clamped to [0, 1]) against the one-context render of all frames
        m_got, m_whole = got.astype(np.float64) / total, whole[..., :3].astype(np.float64) / total
        max_abs = float(np.abs(m_got - m_whole).max())
        mse = float(np.mean((np.clip(m_got, 0, 1) - np.clip(m_whole, 0, 1)) ** 2))
        psnr = float("inf") if mse == 0.0 else 10.0 * np.log10(1.0 / mse)
        if max_abs > 1e-6 * max(1.0, float(m_whole.max())) or psnr < 50.0:
            problems.append("mean radiance: max abs error %g, PSNR %.1f dB" % (max_abs, psnr))
        if not (summed[..., 3] == 1.0).all():
            problems.append("alpha not restored")
        mean = got / np.float32(total)
        disp = np.rint(np.clip(mean, 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
        if not np.array_equal(rgba8[..., :3], disp):
            problems.append("display image is not sum / total_frames")
    dist.barrier()
    dist.destroy_process_group()
    if problems:
        print("FAIL rank %d: %s" % (rank, "; ".join(problems)))
        sys.exit(1)
    print("OK rank %d (%s, world %d, frames %d..%d%s)" % (rank, args.backend, world, first, first + n - 1,
                                                        ", holds the sum" if holds_sum else ""))


if __name__ == "__main__":
    main()
