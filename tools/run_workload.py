"""Launch one benchmark workload a few times through the C ABI (no torch): the program profilers wrap.
usage: run_workload.py [workload] [variant] [launches] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R

WORKLOADS = {"c2_spheres": ("spheres", 1920, 1080, 8), "c3_mesh": ("reference", 1920, 1080, 8),
             "c4_mesh_4k": ("reference", 3840, 2160, 8), "c1_plumbing": ("reference", 640, 360, 4)}
name = sys.argv[1] if len(sys.argv) > 1 else "c2_spheres"
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 1
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 3
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 256
scene_name, w, h, mb = WORKLOADS[name]
scene = R.sphere_scene() if scene_name == "spheres" else R.reference_scene()
fd = R.make_frame_data(w, h, scene, R.BENCH_POSE, 1, mb)
with R.Renderer(w, h, variant=variant) as r:
    r.upload_scene(scene)
    ts = []
    for _ in range(launches):
        r.render_frames(fd, 1, frames)
        ts.append(r.last_render_ms())
    print(name, "variant", variant, "frames", frames, "ms", ["%.3f" % t for t in ts],
          "Gpaths/s %.2f" % (w * h * frames / min(ts) / 1e6))
