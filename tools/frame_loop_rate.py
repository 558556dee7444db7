"""One rtmi_render call per frame, as the reference's frame loop makes them: host time to enqueue a
call, time per frame with the queue kept full, and the last launch's own time, for a few image sizes
and sample counts (reference scene, bench pose, depth 6)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R

scene = R.reference_scene()
for (w, h, rpp) in ((64, 36, 1), (64, 36, 4), (640, 360, 4), (1280, 720, 1), (1280, 720, 4), (1920, 1080, 4), (3840, 2160, 4)):
    fd = R.make_frame_data(w, h, scene, R.BENCH_POSE, rpp, 6)
    with R.Renderer(w, h) as r:
        r.upload_scene(scene)
        for f in range(1, 50):
            fd.frame_index = f
            r.render(fd)
        r.synchronize()
        n = 2000
        t0 = time.perf_counter()
        for f in range(50, 50 + n):
            fd.frame_index = f
            r.render(fd)
        t1 = time.perf_counter()
        r.synchronize()
        t2 = time.perf_counter()
        print("%dx%d, %d samples per pixel: enqueue %.1f us per call, %.1f us per frame, last launch %.1f us" % (
            w, h, rpp, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6, r.last_render_ms() * 1e3), flush=True)
