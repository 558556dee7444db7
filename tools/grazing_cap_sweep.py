"""Cost of the margins given to rays that run almost inside a triangle's plane: one workload, launch time for
a list of rtmi_set_grazing_cap values (the scene is uploaded once; the cap is a kernel argument).
usage: grazing_cap_sweep.py [workload] [frames] [caps...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R

name = sys.argv[1] if len(sys.argv) > 1 else "c5_stress"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16
caps = [float(v) for v in sys.argv[3:]] or [0.0, 2.0 ** -9, 2.0 ** -8, 2.0 ** -7, 2.0 ** -6, 2.0 ** -5, 1.0]
if name == "c5_stress":
    scene, pose = R.scenes.stress_scene()
else:
    scene, pose = R.scenes.stress_scene(5000, (50, 50), 4, extent=5.0)
w, h = 1920, 1080
fd = R.make_frame_data(w, h, scene, pose, 1, 16)
with R.Renderer(w, h, variant=R.abi.VARIANT_AUTO) as r:
    r.upload_scene(scene)
    r.render_frames(fd, 1, frames)   # clocks
    for cap in caps + caps[:1]:
        r.set_grazing_cap(cap)
        ts = []
        for _ in range(3):
            r.render_frames(fd, 1, frames)
            ts.append(r.last_render_ms())
        print("%s frames %d cap %.6f ms min %.2f median %.2f" % (name, frames, cap, min(ts), sorted(ts)[1]), flush=True)
