"""Quick GPU parity + timing probe (development aid)."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import raytracer_amd as R
import oracle_lib as O

def compare(name, W, H, scene, oscene, pose, rpp, mb, first, n, variant):
    fd = R.make_frame_data(W, H, scene, pose, rpp, mb)
    ofd = O.frame_data(W, H, *pose, rpp=rpp, max_bounce=mb, scene=oscene)
    assert bytes(fd) == bytes(ofd)
    ref = O.render(ofd, oscene, first, n)
    with R.Renderer(W, H, variant=variant) as r:
        r.upload_scene(scene)
        r.enable_validation(True)
        r.render_frames(fd, first, n)
        acc = r.read_accum(); ids = r.read_hit_ids(); ch = r.read_chain_hash(); rs = r.read_rng_state(); out = r.read_output_rgba8(); cn = r.counters()
    ok = dict(accum=np.array_equal(acc.view(np.uint32), ref['accum'].view(np.uint32)), ids=np.array_equal(ids, ref['prim_id']),
              chain=np.array_equal(ch, ref['chain_hash']), rng=np.array_equal(rs, ref['rng_state']), rgba8=np.array_equal(out, ref['rgba8']),
              counters=cn == ref['counters'])
    print(name, 'variant', variant, ok)
    if not all(ok.values()):
        d = np.abs(acc - ref['accum']); print('  max abs diff', d.max(), 'n diff px', (d.max(axis=2) > 0).sum(), 'ids diff', (ids != ref['prim_id']).sum(), 'chain diff', (ch != ref['chain_hash']).sum())
        print('  gpu', cn); print('  ref', ref['counters'])
    return all(ok.values())

sc = R.reference_scene(); osc = O.reference_scene()
ssc = R.sphere_scene(); ossc = O.Scene([O.make_sphere(tuple(s.center), s.radius, O.material(tuple(s.material.color), s.material.light, s.material.smoothness)) for s in ssc.spheres])
allok = True
for v in (0, 1):
    allok &= compare('ref/default', 160, 90, sc, osc, R.DEFAULT_POSE, 1, 4, 1, 4, v)
    allok &= compare('ref/bench', 160, 90, sc, osc, R.BENCH_POSE, 1, 8, 1, 8, v)
    allok &= compare('ref/bench rpp4', 100, 50, sc, osc, R.BENCH_POSE, 4, 6, 3, 3, v)
    allok &= compare('spheres/bench', 160, 90, ssc, ossc, R.BENCH_POSE, 1, 8, 1, 8, v)
    allok &= compare('ref/bench rpp3', 70, 33, sc, osc, R.BENCH_POSE, 3, 5, 1, 2, v)
print('ALL OK' if allok else 'MISMATCH')

# timing at full size
for name, scene in (('c2_spheres', ssc), ('c3_mesh', sc)):
    fd = R.make_frame_data(1920, 1080, scene, R.BENCH_POSE, 1, 8)
    for v in (0, 1):
        with R.Renderer(1920, 1080, variant=v) as r:
            r.upload_scene(scene)
            r.render_frames(fd, 1, 16); r.synchronize()
            ts = []
            for it in range(3):
                r.render_frames(fd, 1, 256); ts.append(r.last_render_ms())
            paths = 1920*1080*256
            print(name, 'variant', v, 'ms', ts, 'Gpaths/s', paths/min(ts)/1e6)
