"""Synthetic large scenes: generator code, not data.  The reference's own scenes hold a handful of
primitives (main.cpp:125-185); these exist to exercise the same ClosestHit semantics
(Raytracing.comp:338-388) at sizes where the hierarchy is needed.

Everything is drawn from numpy's MT19937 `RandomState(seed)`, whose uniform stream is frozen, so a
(seed, sizes) pair names one scene everywhere."""
import numpy as np

from . import host


def stress_scene(n_spheres=1_000_000, grid=(1000, 500), n_meshes=16, seed=12345, extent=50.0,
                 radius_range=(0.05, 0.25), glass=False):
    """n_spheres spheres with centres uniform in [-extent, extent]^3 and radii uniform in
    radius_range, plus a displaced grid of grid[0] x grid[1] cells (two triangles each) spanning the
    same square at about y = -0.3 extent, cut into n_meshes strips with one Mesh record each.
    Materials cycle diffuse / mirror / half-smooth; one sphere in 61 emits light.
    glass=True (an extension: the reference has no transmitting material, Raytracing.comp:26-34) makes every fourth
    sphere and every fourth mesh strip transmit with an index of refraction of 1.5; it takes effect only on a
    renderer with the transmission extension enabled, and draws nothing from the random stream, so the
    geometry and the other materials are those of the plain scene.
    Returns (Scene, pose) where pose looks into the cube from outside its -z face."""
    rs = np.random.RandomState(seed)
    e = float(extent)

    sph = np.zeros(n_spheres, dtype=host.SPHERE_DTYPE)
    sph["center"] = rs.uniform(-e, e, size=(n_spheres, 3)).astype(np.float32)
    sph["radius"] = rs.uniform(radius_range[0], radius_range[1], size=n_spheres).astype(np.float32)
    idx = np.arange(n_spheres)
    sph["material"]["color"] = rs.uniform(0.2, 1.0, size=(n_spheres, 3)).astype(np.float32)
    sph["material"]["smoothness"] = np.choose(idx % 3, [0.0, 1.0, 0.5]).astype(np.float32)
    sph["material"]["light"] = np.where(idx % 61 == 7, 2.0, 0.0).astype(np.float32)
    if glass:
        sph["material"]["pad"][:, 0] = np.where(idx % 4 == 3, 1.0, 0.0).astype(np.float32)
        sph["material"]["pad"][:, 1] = 1.5

    nx, nz = int(grid[0]), int(grid[1])
    if nx * nz == 0 or n_meshes == 0:
        return host.Scene.from_records(sph, np.zeros(0, host.TRIANGLE_DTYPE), np.zeros(0, host.MESH_DTYPE)), _pose(e)
    n_meshes = min(n_meshes, nz)
    xs = np.linspace(-e, e, nx + 1)
    zs = np.linspace(-e, e, nz + 1)
    X, Z = np.meshgrid(xs, zs)                      # [nz + 1, nx + 1]
    fx, fz = 6.0 * np.pi / (2 * e), 4.0 * np.pi / (2 * e)
    amp = 0.04 * e
    Y = -0.3 * e + amp * np.sin(fx * X) * np.cos(fz * Z) + 0.1 * amp * rs.uniform(-1.0, 1.0, size=X.shape)
    # smooth shading normals from the analytic part of the height field
    dYdx = amp * fx * np.cos(fx * X) * np.cos(fz * Z)
    dYdz = -amp * fz * np.sin(fx * X) * np.sin(fz * Z)
    N = np.stack([-dYdx, np.ones_like(X), -dYdz], axis=-1)
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    P = np.stack([X, Y, Z], axis=-1).astype(np.float32)
    N = N.astype(np.float32)

    # cell (k, i): corners 00 = (i, k), 10 = (i + 1, k), 01 = (i, k + 1), 11; both triangles wind so
    # that AB x AC points up (RayTriangle culls the other side, Raytracing.comp:293-294)
    p00, p10, p01, p11 = P[:-1, :-1], P[:-1, 1:], P[1:, :-1], P[1:, 1:]
    n00, n10, n01, n11 = N[:-1, :-1], N[:-1, 1:], N[1:, :-1], N[1:, 1:]
    tri = np.zeros((nz, nx, 2), dtype=host.TRIANGLE_DTYPE)
    for name, a, b in (("p1", p00, p00), ("p2", p01, p11), ("p3", p11, p10)):
        tri[name][:, :, 0, :3] = a
        tri[name][:, :, 1, :3] = b
    for name, a, b in (("n1", n00, n00), ("n2", n01, n11), ("n3", n11, n10)):
        tri[name][:, :, 0, :3] = a
        tri[name][:, :, 1, :3] = b

    meshes = np.zeros(n_meshes, dtype=host.MESH_DTYPE)
    bounds = np.linspace(0, nz, n_meshes + 1).astype(int)
    for m in range(n_meshes):
        k0, k1 = bounds[m], bounds[m + 1]
        meshes["start_triangle"][m] = k0 * nx * 2
        meshes["num_triangles"][m] = (k1 - k0) * nx * 2
        verts = P[k0:k1 + 1].reshape(-1, 3)
        # the box LoadModel keeps: componentwise min / max of the vertices (ModelLoader.cpp:96-116)
        meshes["bounding_point1"][m, :3] = verts.min(axis=0)
        meshes["bounding_point2"][m, :3] = verts.max(axis=0)
        meshes["material"]["color"][m] = rs.uniform(0.3, 0.9, size=3).astype(np.float32)
        meshes["material"]["smoothness"][m] = (0.0, 1.0, 0.5)[m % 3]
        if glass and m % 4 == 3:
            meshes["material"]["pad"][m, 0] = 1.0
            meshes["material"]["pad"][m, 1] = 1.5
    return host.Scene.from_records(sph, tri.reshape(-1), meshes), _pose(e)


def _pose(e):
    return ((0.0, 0.3 * e, -1.4 * e), 90.0, -15.0)
