"""Launch shapes (frames x rays per pixel) under the default variant, from stored primary hits: microseconds per
launch for both benchmark scenes at three image sizes.  usage: auto_rule_sweep.py TAG  (TAG labels the lines; run it
with RTMI_LIBRARY pointing at builds of rtmi_api.cpp with -DRTMI_AUTO_COMPACT_FRAMES=1 (always the compacting pair),
=1000000 (always the one-lane-per-pixel kernel) and the shipped library)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R
tag = sys.argv[1]
for scene_name in ("reference", "spheres"):
    scene = R.reference_scene() if scene_name == "reference" else R.sphere_scene()
    for (w, h) in ((1280, 720), (1920, 1080), (3840, 2160)):
        out = []
        for (frames, rpp) in ((1, 8), (1, 16), (2, 4), (2, 8), (3, 4), (4, 2), (4, 4), (6, 2), (8, 1), (8, 4), (16, 1)):
            fd = R.make_frame_data(w, h, scene, R.BENCH_POSE, rpp, 6)
            with R.Renderer(w, h, variant=3) as r:
                r.upload_scene(scene)
                f = 1
                for _ in range(20):
                    r.render_frames(fd, f, frames); f += frames
                r.synchronize()
                n = 300
                t0 = time.perf_counter()
                for _ in range(n):
                    r.render_frames(fd, f, frames); f += frames
                r.synchronize()
                t2 = time.perf_counter()
                out.append("%dx%d:%.0f" % (frames, rpp, (t2 - t0) / n * 1e6))
        print(tag, scene_name, w, h, " ".join(out), flush=True)
