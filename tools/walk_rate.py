"""Rate of the plain (non-resumable) hierarchy walk kernel on random rays inside the stress scene:
how fast is traversal alone, at the occupancy a small kernel gets?  Run under rocprofv3 --kernel-trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
scene, pose = R.scenes.stress_scene()
rs = np.random.RandomState(1)
o = rs.uniform(-50, 50, size=(n, 3)).astype(np.float32)
d = rs.normal(size=(n, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
with R.Renderer(8, 8) as r:
    r.upload_scene(scene)
    for _ in range(3):
        t, prim = r.selftest_closest_hit(scene.n_spheres, scene.n_meshes, o, d)
    print("rays", n, "hit fraction", float((prim != R.abi.MISS_ID).mean()), "mean t of hits", float(t[prim != R.abi.MISS_ID].mean()))
