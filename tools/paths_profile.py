"""Lane-occupancy budget of k_paths from a profiling build (tools/build_variant.sh prof -DRT_PATHS_PROFILE):
RTMI_LIBRARY=scratch/librtmi_prof.so python tools/paths_profile.py [workload] [frames]
Prints, per block of the kernel's loop, how many lanes were active and how often the wave ran it.
With a build made with -DRT_PATHS_PROFILE=2 (tools/build_variant.sh prof2 -DRT_PATHS_PROFILE=2) and a third
argument "meshes", the mesh part of the trace block instead of the sphere candidate blocks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as R

name = sys.argv[1] if len(sys.argv) > 1 else "c2_spheres"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
mesh_mode = len(sys.argv) > 3 and sys.argv[3] == "meshes"
scene_name, w, h, mb = {"c2_spheres": ("spheres", 1920, 1080, 8), "c3_mesh": ("reference", 1920, 1080, 8),
                        "c4_mesh_4k": ("reference", 3840, 2160, 8), "c1_plumbing": ("reference", 640, 360, 4)}[name]
scene = R.sphere_scene() if scene_name == "spheres" else R.reference_scene()
fd = R.make_frame_data(w, h, scene, R.BENCH_POSE, 1, mb)
with R.Renderer(w, h, variant=2) as r:
    r.upload_scene(scene)
    r.reset_counters()
    r.render_frames(fd, 1, frames)
    out = np.zeros(16, dtype=np.uint64)
    r._check(r._lib.rtmi_get_counters(r._ctx, out.ctypes.data, 16))
c = [int(v) for v in out]
trips = c[0]
names = {9: "lanes holding an item", 1: "lanes tracing (sphere loop)", 2: "lanes in hit block", 3: "lanes ending a path",
         4: "lanes evaluating sky", 6: "lanes starting a path", 8: "lanes in new-ray block", 11: "sphere candidate-block lanes (sum over spheres)",
         13: "lanes in literal re-evaluation"}
runs = {10: "trace block", 15: "hit block", 5: "sky block", 7: "start block", 12: "sphere candidate blocks (sum over spheres)", 14: "literal re-evaluation"}
if mesh_mode:
    names = {9: "lanes holding an item", 1: "lanes tracing", 2: "lanes in hit block", 3: "lanes ending a path", 4: "lanes evaluating sky",
             6: "lanes starting a path", 8: "lanes in new-ray block", 11: "rays through a mesh's box gate (sum over meshes)",
             13: "(ray, triangle) tests in the shared triangle loop", 5: "lanes in the unshared triangle loop"}
    runs = {12: "shared mesh block (sum over meshes)", 14: "trips of the shared triangle loop", 15: "cross-lane collect steps (sum of lanes per ray)",
            10: "unshared triangle loop", 7: "start block"}
print("%s, %d frames: %d wave-trips" % (name, frames, trips))
for k, n in names.items():
    print("  %-48s %14d  = %5.1f lanes per trip" % (n, c[k], c[k] / trips))
for k, n in runs.items():
    print("  runs of %-40s %14d  = %5.3f per trip" % (n, c[k], c[k] / trips))
if mesh_mode:
    print("  rays per run of the shared block: %.1f; lanes testing a triangle per trip of its loop: %.1f of 64; loop trips per run: %.2f; "
          "collect steps per run: %.2f" % (c[11] / max(1, c[12] + c[10]), c[13] / max(1, c[14]), c[14] / max(1, c[12]), c[15] / max(1, c[12])))
else:
    print("  candidate block occupancy: %.1f lanes per run" % (c[11] / max(1, c[12])))
