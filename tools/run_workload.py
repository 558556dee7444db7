"""Launch one benchmark workload a few times through the C ABI (no torch): the program profilers wrap.
usage: run_workload.py [workload] [variant] [launches] [frames] [accel]
accel: auto (default), none, bvh"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R

WORKLOADS = {"c2_spheres": ("spheres", 1920, 1080, 8), "c3_mesh": ("reference", 1920, 1080, 8),
             "c4_mesh_4k": ("reference", 3840, 2160, 8), "c1_plumbing": ("reference", 640, 360, 4),
             "c5_stress": ("stress", 1920, 1080, 16), "c5_small": ("stress_small", 1920, 1080, 16),
             "c5_stress_glass": ("stress_glass", 1920, 1080, 16),
             # the reference scene at sizes between C1 and C3 (tuning of the work-list dealing)
             "ref_720p": ("reference", 1280, 720, 8), "ref_540p": ("reference", 960, 540, 8), "ref_1440p": ("reference", 2560, 1440, 8)}
name = sys.argv[1] if len(sys.argv) > 1 else "c2_spheres"
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 2
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 3
frames = int(sys.argv[4]) if len(sys.argv) > 4 else (16 if name.startswith("c5") else 256)
accel = {"auto": R.abi.ACCEL_AUTO, "none": R.abi.ACCEL_NONE, "bvh": R.abi.ACCEL_BVH}[sys.argv[5] if len(sys.argv) > 5 else "auto"]
scene_name, w, h, mb = WORKLOADS[name]
pose = R.BENCH_POSE
t0 = time.perf_counter()
if scene_name == "stress":
    scene, pose = R.scenes.stress_scene()
elif scene_name == "stress_glass":
    scene, pose = R.scenes.stress_scene(glass=True)
elif scene_name == "stress_small":
    scene, pose = R.scenes.stress_scene(5000, (50, 50), 4, extent=5.0)
else:
    scene = R.sphere_scene() if scene_name == "spheres" else R.reference_scene()
t_gen = time.perf_counter() - t0
fd = R.make_frame_data(w, h, scene, pose, 1, mb)
with R.Renderer(w, h, variant=variant) as r:
    r.set_acceleration(accel)
    if scene_name == "stress_glass":
        r.enable_transmission(True)
    t0 = time.perf_counter()
    r.upload_scene(scene)
    t_up = time.perf_counter() - t0
    ts = []
    for _ in range(launches):
        r.render_frames(fd, 1, frames)
        ts.append(r.last_render_ms())
    print(name, "variant", variant, "frames", frames, "accel", r.acceleration(), "ms", ["%.3f" % t for t in ts],
          "Gpaths/s %.3f" % (w * h * frames / min(ts) / 1e6), "scene gen %.2f s, upload %.2f s" % (t_gen, t_up), flush=True)
    if os.environ.get("RTMI_COUNTERS") == "1":
        r.enable_validation(True)
        r.reset_counters()
        r.render_frames(fd, 1, frames)
        print(r.counters())
