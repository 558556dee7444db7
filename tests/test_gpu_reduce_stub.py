"""rtmi_reduce_accum and rtmi_headless --gpus N with more than one rank on a box that has ONE GPU: the
collective library is replaced (RTMI_RCCL_LIBRARY) by tests/rccl_stub, which sums by copy in rank order, so
the grouped reduce, the communicator cache, the device bookkeeping and the resolve run for real.  The
product default never loads the stub: without the variable, two contexts on one GPU are refused."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
STUB = os.path.join(HERE, "rccl_stub", "librccl_stub.so")


def _need_stub():
    if not os.path.exists(STUB):
        pytest.fail("tests/rccl_stub/librccl_stub.so is missing: build it with `make -C tests/rccl_stub` (or __graft_entry__.build())")


def test_contexts_sharing_a_gpu_are_refused_without_the_stub(rtmi):
    lib = rtmi.abi.load()
    if os.environ.get("RTMI_RCCL_LIBRARY"):
        pytest.skip("RTMI_RCCL_LIBRARY is set for the whole run")
    with rtmi.Renderer(32, 18) as a, rtmi.Renderer(32, 18) as b:
        arr = (C.c_void_p * 2)(a._ctx, b._ctx)
        rc = lib.rtmi_reduce_accum(C.cast(arr, C.POINTER(C.c_void_p)), 2, 0)
        assert rc == rtmi.abi.ERR_INVALID_ARGUMENT
        assert b"same GPU" in lib.rtmi_last_error(a._ctx)
        same = (C.c_void_p * 2)(a._ctx, a._ctx)
        assert lib.rtmi_reduce_accum(C.cast(same, C.POINTER(C.c_void_p)), 2, 0) == rtmi.abi.ERR_INVALID_ARGUMENT


def test_in_process_sum_of_three_and_two_ranks_on_one_gpu():
    _need_stub()
    env = dict(os.environ, RTMI_RCCL_LIBRARY=STUB)
    out = subprocess.run([sys.executable, os.path.join(HERE, "reduce_stub_check.py")], env=env, capture_output=True, text=True,
                         timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["stub"] == 1 and rep["shutdown"] == 0
    assert [(r["world"], r["root"]) for r in rep["runs"]] == [(3, 1), (2, 0), (2, 1)]
    for r in rep["runs"]:
        assert r["same_sum"] and r["alpha_one"] and r["other_rank_kept"], r


def test_headless_two_ranks_on_one_gpu(rtmi, tmp_path):
    _need_stub()
    w, h, frames = 64, 36, 7
    exe = os.path.join(rtmi.abi.PKG_DIR, "rtmi_headless")
    mesh_dir = os.path.join(os.path.dirname(rtmi.abi.PKG_DIR), "assets", "meshes")
    pfm = tmp_path / "two.pfm"
    args = [exe, "--size", str(w), str(h), "--meshes", mesh_dir, "--frames", str(frames), "--gpus", "2",
            "--pose", "0", "1.5", "-4.5", "90", "-15", "--pfm", str(pfm)]
    # without the stub the one-GPU box refuses two ranks
    if rtmi.abi.load().rtmi_device_count() < 2:
        plain = subprocess.run(args, capture_output=True, text=True, timeout=120,
                               env={k: v for k, v in os.environ.items() if k != "RTMI_RCCL_LIBRARY"})
        assert plain.returncode != 0 and "only 1 device" in plain.stderr
    out = subprocess.run(args, capture_output=True, text=True, timeout=120, env=dict(os.environ, RTMI_RCCL_LIBRARY=STUB))
    assert out.returncode == 0, out.stderr[-2000:]
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert info["gpus"] == 2 and info["frames"] == frames
    # ranks 0 / 1 render frames 1..4 / 5..7 into cleared images; the sum of the two is what the file holds
    scene = rtmi.reference_scene()
    fd = rtmi.make_frame_data(w, h, scene, rtmi.BENCH_POSE, rays_per_pixel=4, max_bounce=6)
    parts = []
    for first, n in ((1, 4), (5, 3)):
        with rtmi.Renderer(w, h) as r:
            r.upload_scene(scene)
            r.clear_accum()
            r.render_frames(fd, first, n)
            parts.append(r.read_accum())
    mean = (parts[0] + parts[1])[..., :3] / np.float32(frames)
    from test_gpu_headless import _read_pfm
    assert cases.bits_equal(_read_pfm(pfm), mean)
