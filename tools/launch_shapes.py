"""Launch shapes (frames x rays per pixel) through one variant: median milliseconds per launch of ten, both
benchmark scenes at three image sizes.  usage: launch_shapes.py [variant]   (2 = the compacting pair, the default;
1 = the one-lane-per-pixel kernel)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 2
shapes = ((1, 1), (1, 4), (1, 16), (2, 4), (2, 8), (4, 4), (8, 1), (8, 4), (16, 4), (64, 1), (64, 4))
print("# variant %d, us per launch, FRAMESxRAYS_PER_PIXEL:us" % variant)
for scene_name in ("reference", "spheres"):
    scene = R.reference_scene() if scene_name == "reference" else R.sphere_scene()
    for (w, h) in ((1280, 720), (1920, 1080), (3840, 2160)):
        out = []
        for (frames, rpp) in shapes:
            fd = R.make_frame_data(w, h, scene, R.BENCH_POSE, rpp, 6)
            with R.Renderer(w, h, variant=variant) as r:
                r.upload_scene(scene)
                f = 1
                for _ in range(6):
                    r.render_frames(fd, f, frames); f += frames
                r.synchronize()
                ts = []
                for _ in range(10):
                    r.render_frames(fd, f, frames); f += frames
                    r.synchronize()
                    ts.append(r.last_render_ms())
                out.append("%dx%d:%.0f" % (frames, rpp, sorted(ts)[len(ts) // 2] * 1e3))
        print(scene_name, w, h, " ".join(out), flush=True)
