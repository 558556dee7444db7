"""End-to-end runs of raytracer_amd/rtmi_headless (the reference's main() without a window, built on
the C ABI alone): its pictures must be the ones the Python binding renders for the same inputs."""
import json
import os
import subprocess

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

W, H = 64, 36
POSE = ["0", "1.5", "-4.5", "90", "-15"]


def _run(rtmi, *args):
    exe = os.path.join(rtmi.abi.PKG_DIR, "rtmi_headless")
    mesh_dir = os.path.join(os.path.dirname(rtmi.abi.PKG_DIR), "assets", "meshes")
    out = subprocess.run([exe, "--size", str(W), str(H), "--meshes", mesh_dir, *args], check=True, capture_output=True,
                         text=True, timeout=120)
    return json.loads(out.stdout.strip().splitlines()[-1])


def _read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(v) for v in f.readline().split())
        assert float(f.readline()) < 0           # little endian
        return np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)


def _read_ppm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = (int(v) for v in f.readline().split())
        assert int(f.readline()) == 255
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(h, w, 3)


def _python_render(rtmi, scene, first, n, rpp=4, depth=6, seed=None):
    fd = rtmi.make_frame_data(W, H, scene, rtmi.BENCH_POSE, rays_per_pixel=rpp, max_bounce=depth)
    with rtmi.Renderer(W, H) as r:
        r.upload_scene(scene)
        if seed is not None:
            r.write_accum(seed)
        r.render_frames(fd, first, n)
        return r.read_accum(), r.read_output_rgba8()


def test_frame_loop_and_batch_write_the_same_pictures(rtmi, tmp_path):
    pfm, ppm = tmp_path / "a.pfm", tmp_path / "a.ppm"
    info = _run(rtmi, "--frames", "5", "--pose", *POSE, "--pfm", str(pfm), "--ppm", str(ppm))
    assert info["frames"] == 5 and info["frame_index"] == 5 and info["rays_per_pixel"] == 4 and info["max_bounce"] == 6
    accum, rgba = _python_render(rtmi, rtmi.reference_scene(), 1, 5)
    mean = accum[..., :3] / np.float32(5)
    assert cases.bits_equal(_read_pfm(pfm), mean)            # rows bottom-up in both
    # the PPM is the display image flipped upright and sRGB encoded
    lin = rgba[::-1, :, :3].astype(np.float64) / 255.0
    srgb = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * np.power(lin, 1 / 2.4) - 0.055)
    want = np.floor(np.clip(srgb, 0, 1) * 255.0 + 0.5).astype(np.int32)
    assert np.abs(_read_ppm(ppm).astype(np.int32) - want).max() <= 1
    pfm2, exr = tmp_path / "b.pfm", tmp_path / "b.exr"
    _run(rtmi, "--frames", "5", "--batch", "--pose", *POSE, "--pfm", str(pfm2), "--exr", str(exr))
    assert cases.bits_equal(_read_pfm(pfm2), mean)
    # the EXR holds the same mean radiance, top row first, as planes B, G, R
    from test_host import _read_exr
    _, planes = _read_exr(exr.read_bytes())
    assert cases.bits_equal(np.stack([planes["R"], planes["G"], planes["B"]], axis=-1), mean[::-1])


def test_scene_file_script_and_snapshots(rtmi, tmp_path):
    scene = rtmi.sphere_scene()
    scene_file, snap, pfm = tmp_path / "s.txt", tmp_path / "acc.snap", tmp_path / "c.pfm"
    scene.save(scene_file)
    script = tmp_path / "cam.txt"
    script.write_text("# frames dt keys mouse_x mouse_y button2 reset\n3 0.016 - 0 0 0 0\n")
    _run(rtmi, "--scene", str(scene_file), "--script", str(script), "--pose", *POSE, "--checkpoint", str(snap))
    info = _run(rtmi, "--scene", str(scene_file), "--resume", str(snap), "--frames", "4", "--batch", "--pose", *POSE,
                "--pfm", str(pfm))
    assert info["frame_index"] == 7
    accum, _ = _python_render(rtmi, scene, 1, 7)
    assert cases.bits_equal(_read_pfm(pfm), accum[..., :3] / np.float32(7))
    # a script whose camera moves restarts the sum: the last run of still frames is all that remains
    script.write_text("2 0.016 - 0 0 0 0\n1 0.25 W 0 0 0 0\n3 0.016 - 0 0 0 0\n")
    info = _run(rtmi, "--scene", str(scene_file), "--script", str(script), "--pose", *POSE)
    assert info["frames"] == 6 and info["frame_index"] == 4          # 1, 2, (moved) 1, 2, 3, 4


def test_gpus_option(rtmi, tmp_path):
    """--gpus N splits the frames over N GPUs and sums the images with RCCL (rtmi_reduce_accum).  One GPU:
    the same picture as --batch, bit for bit.  More GPUs than the machine has: a clear message.  With two
    or more GPUs: the two-GPU picture against the one-GPU picture within the tolerance of a re-associated sum."""
    exe = os.path.join(rtmi.abi.PKG_DIR, "rtmi_headless")
    have = rtmi.abi.load().rtmi_device_count()
    assert have >= 1
    a, b = tmp_path / "a.pfm", tmp_path / "b.pfm"
    one = _run(rtmi, "--frames", "6", "--rpp", "1", "--bounces", "8", "--pose", *POSE, "--batch", "--pfm", str(a))
    grp = _run(rtmi, "--frames", "6", "--rpp", "1", "--bounces", "8", "--pose", *POSE, "--gpus", "1", "--pfm", str(b))
    assert (one["gpus"], grp["gpus"]) == (1, 1) and grp["frame_index"] == 6
    assert cases.bits_equal(_read_pfm(a), _read_pfm(b))
    out = subprocess.run([exe, "--size", str(W), str(H), "--gpus", str(have + 1)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "only %d device" % have in out.stderr
    if have >= 2:
        c = tmp_path / "c.pfm"
        two = _run(rtmi, "--frames", "6", "--rpp", "1", "--bounces", "8", "--pose", *POSE, "--gpus", "2", "--pfm", str(c))
        assert two["gpus"] == 2
        x, y = _read_pfm(a), _read_pfm(c)
        assert np.allclose(x, y, rtol=0, atol=6 * 4e-7 * max(1.0, float(x.max())))


def test_reduce_accum_argument_checks(rtmi):
    lib = rtmi.abi.load()
    import ctypes as C
    with rtmi.Renderer(16, 8) as r1, rtmi.Renderer(16, 8) as r2, rtmi.Renderer(8, 8) as r3:
        one = (C.c_void_p * 1)(r1._ctx)
        assert lib.rtmi_reduce_accum(one, 1, 0) == 0                 # a sum over one image
        assert lib.rtmi_reduce_accum(one, 1, 1) == rtmi.abi.ERR_INVALID_ARGUMENT
        same_gpu = (C.c_void_p * 2)(r1._ctx, r2._ctx)
        assert lib.rtmi_reduce_accum(same_gpu, 2, 0) == rtmi.abi.ERR_INVALID_ARGUMENT
        assert b"same GPU" in lib.rtmi_last_error(r1._ctx)
        other_size = (C.c_void_p * 2)(r1._ctx, r3._ctx)
        assert lib.rtmi_reduce_accum(other_size, 2, 0) == rtmi.abi.ERR_INVALID_ARGUMENT
