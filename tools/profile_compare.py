#!/usr/bin/env python3
"""Strict against fused arithmetic profile on the GPU: for each workload, how many pixels get another
primary hit, in how many pixels some path takes another course (id-chain hash), PSNR and largest
difference of the mean radiance images, and the kernel time of both.  One JSON document on stdout.

    python tools/profile_compare.py [c2_spheres c3_mesh ...] > profiles/r02_strict_vs_fma.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import raytracer_amd as R  # noqa: E402
from bench import WORKLOADS, build_scene  # noqa: E402


def run(workload, frames):
    scene_name, w, h, mb = WORKLOADS[workload]
    scene, pose = build_scene(R, scene_name)
    fd = R.make_frame_data(w, h, scene, pose, rays_per_pixel=1, max_bounce=mb)
    variant = 4 if workload.startswith("c5") else 2
    out = {}
    for profile in ("strict", "fma"):
        with R.Renderer(w, h, variant=variant, profile=profile) as r:
            r.upload_scene(scene)
            r.enable_validation(True)
            r.render_frames(fd, 1, frames)
            ids, chain, accum, counters = r.read_hit_ids(), r.read_chain_hash(), r.read_accum(), r.counters()
            r.enable_validation(False)
            r.render_frames(fd, 1, frames)
            r.synchronize()
            r.timing_reset(True)
            for _ in range(3):
                r.render_frames(fd, 1, frames)
            r.synchronize()
            ms, launches = r.timing_read()
            r.timing_reset(False)
            out[profile] = dict(ids=ids, chain=chain, accum=accum, segments=counters["n_segments"], ms=ms / max(1, launches))
    a, b = out["strict"], out["fma"]
    ma = np.clip(a["accum"][..., :3] / np.float32(frames), 0.0, 1.0).astype(np.float64)
    mb_ = np.clip(b["accum"][..., :3] / np.float32(frames), 0.0, 1.0).astype(np.float64)
    mse = float(np.mean((ma - mb_) ** 2))
    return {
        "workload": workload, "width": w, "height": h, "frames": frames, "max_bounce": mb,
        "pixels": w * h,
        "pixels_with_another_primary_hit": int((a["ids"] != b["ids"]).sum()),
        "pixels_where_some_path_differs": int((a["chain"] != b["chain"]).sum()),
        "segments_strict": a["segments"], "segments_fma": b["segments"],
        "psnr_db_mean_radiance_clamped": (10.0 * np.log10(1.0 / mse)) if mse > 0 else None,
        "max_abs_difference_mean_radiance": float(np.abs(ma - mb_).max()),
        "kernel_ms_strict": a["ms"], "kernel_ms_fma": b["ms"], "speedup": a["ms"] / b["ms"],
    }


def main():
    names = sys.argv[1:] or ["c2_spheres", "c3_mesh"]
    frames = {"c5_stress": 4, "c4_mesh_4k": 64}
    print(json.dumps({"source": "tools/profile_compare.py", "results": [run(n, frames.get(n, 256)) for n in names]}, indent=1))


if __name__ == "__main__":
    main()
