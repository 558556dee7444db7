"""Randomised parity campaign: many small generated scenes, random camera poses, sample counts, bounce
limits, frame ranges and kernel variants, each rendered on the GPU with validation on and compared
bit for bit (accumulation image, display image, primitive ids, id-chain hash, RNG state) with the CPU
oracle.  Not part of the test suite (it runs as long as it is told to).
usage: fuzz_parity.py [n_cases] [first_seed]
RTMI_FUZZ_GLASS=1: every fourth sphere and mesh of the generated scenes transmits, and both sides run with the
transmitting-material extension switched on (the kernels against the oracle's statement of the same rule)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracer_amd as R
import oracle_lib as O

glass = os.environ.get("RTMI_FUZZ_GLASS") == "1"
if glass:
    O.lib().orc_set_transmission(1)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
for case in range(n_cases):
    rs = np.random.RandomState(seed0 + case)
    n_sph = int(rs.choice([0, 1, 2, 7, 30, 90, 300, 700]))
    n_mesh = int(rs.choice([0, 1, 2, 5]))
    grid = (int(rs.randint(1, 9)), int(rs.randint(max(1, n_mesh), 12))) if n_mesh else (0, 0)
    extent = float(rs.uniform(1.5, 6.0))
    rad = float(rs.uniform(0.1, 0.6))
    scene, pose = R.scenes.stress_scene(n_sph, grid, n_mesh, seed=seed0 + case, extent=extent, radius_range=(0.3 * rad, rad), glass=glass)
    pos = tuple(float(v) for v in rs.uniform(-1.2 * extent, 1.2 * extent, size=3))
    yaw, pitch = float(rs.uniform(-180, 180)), float(rs.uniform(-80, 80))
    if rs.rand() < 0.5:
        pos, yaw, pitch = pose                      # the generator's own view of the scene
    w, h = int(rs.choice([8, 23, 40, 64, 128])), int(rs.choice([8, 17, 36, 72]))
    rpp, mb = int(rs.choice([1, 1, 2, 3, 5])), int(rs.choice([0, 1, 2, 3, 6, 16]))
    first, n = int(rs.choice([1, 1, 2, 9])), int(rs.choice([1, 2, 5, 11, 40]))
    accel = int(rs.choice([R.abi.ACCEL_NONE, R.abi.ACCEL_BVH]))
    variant = int(rs.choice([0, 1, 2, 3, 4]))        # 3 chooses per launch and keeps primary hits between launches
    # a device taken to be small sends these small images through every way of dealing work to waves (neighbouring
    # or spread pixels, 64 .. 8 to a wave, few persistent waves)
    cus = int(rs.choice([0, 0, 1, 1, 2, 4, 16]))
    osc = O.Scene.from_product(scene)
    fd = R.make_frame_data(w, h, scene, pose=(pos, yaw, pitch), rays_per_pixel=rpp, max_bounce=mb)
    ofd = O.frame_data(w, h, pos=pos, yaw=yaw, pitch=pitch, rpp=rpp, max_bounce=mb, scene=osc)
    seed_acc = None
    if first != 1:
        seed_acc = rs.uniform(0.0, 3.0, size=(h, w, 4)).astype(np.float32)
        seed_acc[..., 3] = 1.0
    want = O.render(ofd, osc, first, n, accum=seed_acc)
    with R.Renderer(w, h, variant=variant) as r:
        r.set_acceleration(accel)
        r.enable_transmission(glass)
        r.assume_compute_units(cus)
        r.upload_scene(scene)
        r.enable_validation(True)
        if seed_acc is not None:
            r.write_accum(seed_acc)
        # in one launch, or in two that continue each other
        n1 = n if n < 2 or rs.rand() < 0.5 else int(rs.randint(1, n))
        r.render_frames(fd, first, n1)
        if n1 < n:
            r.render_frames(fd, first + n1, n - n1)
        got = dict(accum=r.read_accum(), rgba8=r.read_output_rgba8(), prim_id=r.read_hit_ids(),
                   chain_hash=r.read_chain_hash(), rng_state=r.read_rng_state())
    if n1 < n:
        del got["chain_hash"]           # the hash describes one launch's frames, not a running total
    diffs = [k for k in got if not np.array_equal(got[k].view(np.uint8), np.asarray(want[k]).view(np.uint8))]
    tag = "spheres %d meshes %d tris %d, %dx%d rpp %d depth %d frames %d..%d accel %d variant %d" % (
        n_sph, n_mesh, scene.n_triangles, w, h, rpp, mb, first, first + n - 1, accel, variant)
    if diffs:
        bad += 1
        print("MISMATCH seed %d: %s: %s" % (seed0 + case, tag, diffs), flush=True)
    elif case % 25 == 0:
        print("ok seed %d: %s (hit pixels %d)" % (seed0 + case, tag, int((want["prim_id"] != O.MISS_ID).sum())), flush=True)
print("%d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
