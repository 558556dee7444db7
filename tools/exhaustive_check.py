import sys
sys.path.insert(0,'/root/repo')
import raytracer_amd as R
with R.Renderer(8,8) as r:
    for op,c in ((0,1.0),(1,1.0),(2,0.4),(2,0.01),(2,3.0),(2,0.1)):
        import numpy as np
        print(op,c,[hex(v) for v in r.selftest_exhaustive(op,float(np.float32(c)))],flush=True)
