"""N > 1 path on the CPU: frame-range sharding + one sum collective (gloo, world size 2 and 3).
The per-rank renderer is the CPU oracle standing in for the GPU library; what is under test is the
sharding / reduction logic of raytracer_amd.distributed."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases


def test_split_frames_partitions_exactly():
    from raytracer_amd.distributed import split_frames
    for total in (0, 1, 7, 256, 4096):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                first, n = split_frames(total, world, r, first_frame=5)
                got.extend(range(first, first + n))
            assert got == list(range(5, 5 + total))
            sizes = [split_frames(total, world, r)[1] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert split_frames(4096, 8, 3) == (3 * 512 + 1, 512)
    with pytest.raises(ValueError):
        split_frames(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case_name, total_frames, dst, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as O
        from raytracer_amd.distributed import render_sharded
        case = cases.CASES[case_name]
        sc = cases.oracle_scene(case)
        fd = cases.oracle_frame_data(case, sc)
        w, h = case["size"]
        accum = torch.full((h, w, 4), 7.0, dtype=torch.float32)   # junk: must be cleared or overwritten

        def render_range(first, n):
            r = O.render(fd, sc, first, n, accum=accum.numpy(), n_threads=1)
            accum.copy_(torch.from_numpy(r["accum"]))

        first, n = render_sharded(render_range, accum, total_frames, 1, None, dst)
        np.save(os.path.join(out_dir, "accum_%d.npy" % rank), accum.numpy())
        np.save(os.path.join(out_dir, "range_%d.npy" % rank), np.array([first, n]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total,dst", [(2, 6, None), (2, 5, 0), (3, 2, None)])
def test_sharded_render_matches_single_process(oracle, tmp_path, world, total, dst):
    name = "c3_mesh_bench_pose"
    case = dict(cases.CASES[name], first=1, n=total)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, name, total, dst, str(tmp_path)), nprocs=world, join=True)
    ref = cases.run_oracle(case, n_threads=2)
    ranges = [tuple(np.load(tmp_path / ("range_%d.npy" % r))) for r in range(world)]
    covered = [f for first, n in ranges for f in range(first, first + n)]
    assert covered == list(range(1, total + 1))
    readers = range(world) if dst is None else [dst]
    for r in readers:
        acc = np.load(tmp_path / ("accum_%d.npy" % r))
        # float sums are re-associated (per-rank partial sums, then the collective): a few ulp per frame
        assert np.allclose(acc[..., :3], ref["accum"][..., :3], rtol=0, atol=total * 4e-7 * max(1.0, ref["accum"].max()))
        # the stated tolerance of the mean radiance at equal samples per pixel: per-pixel maximum absolute error
        # of the mean below 1e-6 of the largest value, and PSNR (peak 1, values clamped to [0, 1]) of at least
        # 50 dB; what a re-associated float sum of a few images does is far inside both
        mean, want = acc[..., :3].astype(np.float64) / total, ref["accum"][..., :3].astype(np.float64) / total
        assert np.abs(mean - want).max() <= 1e-6 * max(1.0, float(want.max()))
        mse = float(np.mean((np.clip(mean, 0, 1) - np.clip(want, 0, 1)) ** 2))
        assert mse == 0.0 or 10.0 * np.log10(1.0 / mse) >= 50.0
        n_rendering = sum(1 for _, n in ranges if n > 0)
        assert (acc[..., 3] == float(n_rendering)).all()
    if world == 2 and dst is None:
        a0, a1 = np.load(tmp_path / "accum_0.npy"), np.load(tmp_path / "accum_1.npy")
        assert np.array_equal(a0, a1)
    # the partial images themselves are exact: rank sums equal the oracle on the same frame range
    if dst is None and world == 2 and total == 6:
        part = cases.run_oracle(dict(case, first=4, n=3), n_threads=1)   # second rank's range, from zero
        whole = np.load(tmp_path / "accum_0.npy")
        first_half = cases.run_oracle(dict(case, first=1, n=3), n_threads=1)
        assert np.array_equal((first_half["accum"] + part["accum"])[..., :3].view(np.uint32), whole[..., :3].view(np.uint32))
