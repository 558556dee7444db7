"""Where does the hierarchy overtake the exhaustive loops?  Times both on random sphere scenes of
growing size (same density), 1080p, depth 8."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R

w, h, frames = 1920, 1080, 8
for n in (16, 32, 64, 128, 256, 512, 1024):
    extent = 1.5 * n ** (1.0 / 3.0)
    scene, pose = R.scenes.stress_scene(n, (0, 0), 0, seed=7, extent=extent, radius_range=(0.3, 0.7))
    fd = R.make_frame_data(w, h, scene, pose, 1, 8)
    row = []
    for accel in (R.abi.ACCEL_NONE, R.abi.ACCEL_BVH):
        with R.Renderer(w, h) as r:
            r.set_acceleration(accel)
            r.upload_scene(scene)
            ts = []
            for _ in range(3):
                r.render_frames(fd, 1, frames)
                ts.append(r.last_render_ms())
            row.append(min(ts))
    print("spheres %5d  exhaustive %8.3f ms  hierarchy %8.3f ms" % (n, row[0], row[1]), flush=True)
