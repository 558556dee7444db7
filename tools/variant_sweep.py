"""Time per launch of each kernel variant against frames per launch (1080p, bench pose)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R

w, h = 1920, 1080
for name, scene, rpp, depth in (("spheres rpp1 d8", R.sphere_scene(), 1, 8), ("reference rpp1 d8", R.reference_scene(), 1, 8),
                                ("reference rpp4 d6", R.reference_scene(), 4, 6)):
    fd = R.make_frame_data(w, h, scene, R.BENCH_POSE, rpp, depth)
    for frames in (1, 2, 4, 8, 16, 64):
        row = []
        for variant in (0, 1, 2):
            with R.Renderer(w, h, variant=variant) as r:
                r.upload_scene(scene)
                ts = []
                for _ in range(6):
                    r.render_frames(fd, 1, frames)
                    ts.append(r.last_render_ms())
                row.append(min(ts))
        print("%-18s frames %3d: megakernel %7.3f  regenerate %7.3f  compact %7.3f ms" % (name, frames, *row), flush=True)
