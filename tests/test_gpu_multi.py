"""The N > 1 path over the HIP library: ShardedRenderer with two ranks, and bench.py launching its own
ranks.  On a box with two GPUs the ranks use RCCL; the one-GPU rehearsal (both ranks on GPU 0, gloo)
runs everywhere and keeps the same code from rotting."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _device_count():
    import torch
    return torch.cuda.device_count()


def _run_ranks(world, extra):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "sharded_check.py")] + extra, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=280)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d: %s" % (rank, out[-3000:])
        assert "OK rank %d" % rank in out


@pytest.mark.parametrize("dst", [-1, 0])
def test_two_ranks_on_one_gpu_rehearsal(dst):
    _run_ranks(2, ["--backend", "gloo", "--dst", str(dst)])


def test_three_ranks_on_one_gpu_rehearsal_mesh_scene():
    _run_ranks(3, ["--backend", "gloo", "--dst", "0", "--scene", "reference", "--frames", "10", "--size", "96x54"])


@pytest.mark.parametrize("dst", [-1, 0])
def test_two_ranks_rccl(dst):
    if _device_count() < 2:
        pytest.skip("needs two GPUs")
    _run_ranks(2, ["--backend", "nccl", "--dst", str(dst)])


def _bench(args, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=280)


def test_bench_starts_its_own_ranks_rehearsal():
    """`python bench.py --gpus 2` as the driver invokes it, with both ranks on GPU 0 over gloo."""
    out = _bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--workload", "c1_plumbing", "--no-cpu-baseline"],
                 {"RTMI_BENCH_REHEARSE": "1"})
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 4 and res["value"] > 0
    assert res["config"]["total_frames"] == 2 * res["config"]["frames_per_gpu"]
    # what makes a scaling loss attributable: per-step time of the collective and the render kernels over the ranks
    assert res["reduce_ms"]["max_over_ranks"] >= res["reduce_ms"]["min_over_ranks"] >= 0.0
    assert res["kernel_ms_over_ranks"]["max"] >= res["kernel_ms_over_ranks"]["min"] > 0.0
    assert res["reduce_payload_bytes"] == 640 * 360 * 16


@pytest.mark.parametrize("extra", [[], ["--no-pipeline"], ["--reduce", "allreduce"]])
def test_bench_rehearsal_full_size_image(extra):
    """The default workload's 33 MB image through the N > 1 bench path, six steps: alternating contexts (the default),
    one context, and the all-reduce."""
    out = _bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--frames", "32", "--no-cpu-baseline"] + extra,
                 {"RTMI_BENCH_REHEARSE": "1"})
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["reduce_payload_bytes"] == 1920 * 1080 * 16
    assert ("alternate" in res["config"]["parallelism"]) == (extra != ["--no-pipeline"])


def test_bench_refuses_more_gpus_than_there_are():
    n = _device_count()
    out = _bench(["--gpus", str(n + 1), "--steps", "1", "--warmup", "0"])
    assert out.returncode != 0
    assert "only %d device" % n in out.stderr
    assert "usage" not in out.stderr.lower()


def test_bench_two_gpus_rccl():
    if _device_count() < 2:
        pytest.skip("needs two GPUs")
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "c1_plumbing"])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 2
