"""Where a wave of k_paths spends its cycles, from a stamped build (tools/build_variant.sh stamps -DRT_PATHS_STAMPS):
RTMI_LIBRARY=scratch/librtmi_stamps.so python tools/paths_stamps.py [workload] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as R
name = sys.argv[1] if len(sys.argv) > 1 else "c2_spheres"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
scene_name, w, h, mb = {"c2_spheres": ("spheres", 1920, 1080, 8), "c3_mesh": ("reference", 1920, 1080, 8)}[name]
scene = R.sphere_scene() if scene_name == "spheres" else R.reference_scene()
fd = R.make_frame_data(w, h, scene, R.BENCH_POSE, 1, mb)
with R.Renderer(w, h, variant=2) as r:
    r.upload_scene(scene)
    r.render_frames(fd, 1, frames)
    r.reset_counters()
    r.render_frames(fd, 1, frames)
    ms = r.last_render_ms()
    out = np.zeros(16, dtype=np.uint64)
    r._check(r._lib.rtmi_get_counters(r._ctx, out.ctypes.data, 16))
c = [int(v) for v in out[:8]]
tot = sum(c)
labels = ["trace (sphere loop, winner)", "hit block", "sky block", "end of path, ring, fetch", "start block (cross-lane reads)",
          "new-ray block", "reduce", "loop control"]
print("%s %d frames, %.3f ms with stamps; wave-cycles by block (s_memtime ticks, 100 MHz or shader clock):" % (name, frames, ms))
for l, v in zip(labels, c):
    print("  %-34s %16d  %5.1f %%" % (l, v, 100.0 * v / tot))
