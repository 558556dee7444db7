"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# ORACLE_LIBRARY selects another build of the same file (the sanitizer build, oracle/Makefile target asan)
LIB_PATH = os.environ.get("ORACLE_LIBRARY") or os.path.join(ORACLE_DIR, "liboracle.so")
# The fused arithmetic profile is a second build of the same file; RTMI_PROFILE=fma makes it the default
# of lib() / render() / closest_hit(), as it makes "fma" the default profile of raytracer_amd.Renderer.
FMA_LIB_PATH = os.path.join(ORACLE_DIR, "liboracle_fma.so")
DEFAULT_PROFILE = os.environ.get("RTMI_PROFILE") or "strict"

N_COUNTERS = 16
COUNTER_NAMES = [
    "n_sphere_tests", "n_sphere_disc_ge0", "n_sphere_wins", "n_box_tests", "n_tri_tests",
    "n_tri_wins", "n_hit_bounces", "n_env_evals", "n_pixel_frames", "n_paths", "n_segments",
]
MISS_ID = 0xFFFFFFFF


class Material(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("light", C.c_float), ("smoothness", C.c_float),
                ("pad", C.c_float * 3)]


class Sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("material", Material)]


class Triangle(C.Structure):
    _fields_ = [("p1", C.c_float * 4), ("p2", C.c_float * 4), ("p3", C.c_float * 4),
                ("n1", C.c_float * 4), ("n2", C.c_float * 4), ("n3", C.c_float * 4)]


class Mesh(C.Structure):
    _fields_ = [("start_triangle", C.c_uint32), ("num_triangles", C.c_uint32),
                ("pad", C.c_float * 2), ("bound_lo", C.c_float * 4), ("bound_hi", C.c_float * 4),
                ("material", Material)]


class FrameData(C.Structure):
    _fields_ = [("inv_proj", C.c_float * 16), ("inv_view", C.c_float * 16),
                ("cam_pos", C.c_float * 4), ("cam_dir", C.c_float * 4), ("window", C.c_float * 2),
                ("rays_per_pixel", C.c_uint32), ("max_bounce", C.c_uint32),
                ("sky_horizon", C.c_float * 4), ("sky_zenith", C.c_float * 4),
                ("sun_dir", C.c_float * 4), ("ground", C.c_float * 4),
                ("sun_focus", C.c_float), ("sun_intensity", C.c_float),
                ("frame_index", C.c_uint32), ("sphere_count", C.c_uint32),
                ("mesh_count", C.c_uint32)]


def build(force=False):
    if force or any(not os.path.exists(path) or any(
            os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(path)
            for f in ("rt_oracle.c", "rt_oracle.h", "Makefile")) for path in (LIB_PATH, FMA_LIB_PATH)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_libs = {}


def lib(profile=None):
    """The oracle library of an arithmetic profile ("strict" or "fma"; None = DEFAULT_PROFILE)."""
    profile = profile or DEFAULT_PROFILE
    if profile in _libs:
        return _libs[profile]
    if profile not in ("strict", "fma"):
        raise ValueError("unknown arithmetic profile %r" % (profile,))
    path = FMA_LIB_PATH if profile == "fma" else LIB_PATH
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    f3 = C.POINTER(C.c_float)
    L.orc_sizeof.restype = C.c_size_t
    L.orc_sizeof.argtypes = [C.c_int]
    L.orc_frame_data_offset.restype = C.c_size_t
    L.orc_frame_data_offset.argtypes = [C.c_int]
    L.orc_rng_next.restype = C.c_float
    L.orc_rng_next.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.orc_pixel_seed.restype = C.c_uint32
    L.orc_pixel_seed.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.c_uint32]
    L.orc_ray_sphere.restype = C.c_int
    L.orc_ray_sphere.argtypes = [f3, f3, C.POINTER(Sphere), f3, f3, f3]
    L.orc_ray_triangle.restype = C.c_int
    L.orc_ray_triangle.argtypes = [f3, f3, C.POINTER(Triangle), f3, f3, f3]
    L.orc_ray_box.restype = C.c_int
    L.orc_ray_box.argtypes = [f3, f3, f3, f3]
    L.orc_set_transmission.restype = None
    L.orc_set_transmission.argtypes = [C.c_int]
    L.orc_env_light.restype = None
    L.orc_env_light.argtypes = [C.POINTER(FrameData), f3, f3]
    for name in ("orc_log2", "orc_exp2"):
        getattr(L, name).restype = C.c_float
        getattr(L, name).argtypes = [C.c_float]
    L.orc_pow.restype = C.c_float
    L.orc_pow.argtypes = [C.c_float, C.c_float]
    L.orc_primary_ray.restype = None
    L.orc_primary_ray.argtypes = [C.POINTER(FrameData), C.c_uint32, C.c_uint32, f3, f3]
    L.orc_perspective_fov_lh_zo.restype = None
    L.orc_perspective_fov_lh_zo.argtypes = [C.c_float] * 5 + [f3]
    L.orc_look_at_lh.restype = None
    L.orc_look_at_lh.argtypes = [f3, f3, f3, f3]
    L.orc_mat4_inverse.restype = None
    L.orc_mat4_inverse.argtypes = [f3, f3]
    L.orc_camera_forward.restype = None
    L.orc_camera_forward.argtypes = [C.c_float, C.c_float, f3]
    L.orc_camera_pose.restype = None
    L.orc_camera_pose.argtypes = [C.POINTER(FrameData), f3, C.c_float, C.c_float, C.c_uint32, C.c_uint32]
    L.orc_load_model.restype = C.c_int
    L.orc_load_model.argtypes = [C.c_char_p, C.POINTER(Triangle), C.POINTER(C.c_uint32), C.c_uint32,
                                 C.POINTER(Mesh), C.POINTER(C.c_uint32), C.c_uint32,
                                 C.POINTER(Material), f3, f3, C.c_int]
    L.orc_default_frame_constants.restype = None
    L.orc_default_frame_constants.argtypes = [C.POINTER(FrameData)]
    L.orc_render.restype = C.c_int
    L.orc_render.argtypes = [C.POINTER(FrameData), C.POINTER(Sphere), C.POINTER(Triangle), C.c_uint32,
                             C.POINTER(Mesh), C.c_uint32, C.c_uint32,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_int]
    L.orc_render_rows.restype = C.c_int
    L.orc_render_rows.argtypes = L.orc_render.argtypes + [C.c_uint32, C.c_uint32]
    L.orc_closest_hit.restype = None
    L.orc_closest_hit.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(Sphere), C.POINTER(Triangle), C.POINTER(Mesh),
                                  C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.orc_profile.restype = C.c_int
    L.orc_profile.argtypes = []
    if L.orc_profile() != (1 if profile == "fma" else 0):
        raise RuntimeError("%s was not built for the %s profile" % (path, profile))
    _libs[profile] = L
    return L


def closest_hit(scene, origins, directions, sphere_count=None, mesh_count=None, profile=None):
    """(t, prim_id) per ray, rays as [n, 3] float32 arrays."""
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
    t = np.empty(len(o), dtype=np.float32)
    prim = np.empty(len(o), dtype=np.uint32)
    lib(profile).orc_closest_hit(scene.n_spheres if sphere_count is None else sphere_count,
                          scene.n_meshes.value if mesh_count is None else mesh_count,
                          scene.spheres, scene.tris, scene.meshes, o.ctypes.data, d.ctypes.data, len(o),
                          t.ctypes.data, prim.ctypes.data)
    return t, prim


def fl3(*v):
    return (C.c_float * 3)(*v)


def material(color, light, smoothness):
    m = Material()
    m.color[:] = color
    m.light = light
    m.smoothness = smoothness
    return m


class Scene:
    """Host-side scene in the reference's AoS records (oracle's own definitions)."""

    def __init__(self, spheres=(), tri_cap=1 << 16, mesh_cap=1 << 10):
        self.spheres = (Sphere * max(1, len(spheres)))(*spheres)
        self.n_spheres = len(spheres)
        self.tris = (Triangle * tri_cap)()
        self.n_tris = C.c_uint32(0)
        self.tri_cap = tri_cap
        self.meshes = (Mesh * mesh_cap)()
        self.n_meshes = C.c_uint32(0)
        self.mesh_cap = mesh_cap

    @classmethod
    def from_product(cls, product_scene):
        """Copy of a product-side scene: both sides hold the reference's record layouts
        (RayTracingStructs.h:4-60), so the bytes carry over."""
        ns, nt, nm = product_scene.n_spheres, product_scene.n_triangles, product_scene.n_meshes
        sc = cls((), tri_cap=max(1, nt), mesh_cap=max(1, nm))
        sc.spheres = (Sphere * max(1, ns))()
        sc.n_spheres = ns
        C.memmove(sc.spheres, product_scene.sphere_array(), ns * C.sizeof(Sphere))
        C.memmove(sc.tris, product_scene.triangles, nt * C.sizeof(Triangle))
        C.memmove(sc.meshes, product_scene.meshes, nm * C.sizeof(Mesh))
        sc.n_tris.value = nt
        sc.n_meshes.value = nm
        return sc

    def load_model(self, path, mat, translate=(0, 0, 0), scale=(1, 1, 1), quad_rule=0):
        rc = lib().orc_load_model(path.encode(), self.tris, C.byref(self.n_tris), self.tri_cap,
                                  self.meshes, C.byref(self.n_meshes), self.mesh_cap,
                                  C.byref(mat), fl3(*translate), fl3(*scale), quad_rule)
        if rc < 0:
            raise RuntimeError("orc_load_model failed for %s" % path)
        return rc


def make_sphere(center, radius, mat):
    s = Sphere()
    s.center[:] = center
    s.radius = radius
    s.material = mat
    return s


MESH_DIR = os.path.join(ROOT, "assets", "meshes")


def reference_scene(quad_rule=0):
    """main.cpp:161-185: one sphere, plane.obj scaled (2,1,2), cube.obj translated (-1,1,0)."""
    sc = Scene([make_sphere((1, 1, 0), 0.5, material((1, 1, 1), 0.0, 0.1))])
    sc.load_model(os.path.join(MESH_DIR, "plane.obj"), material((1, 1, 1), 0.0, 0.8),
                  (0, 0, 0), (2, 1, 2), quad_rule)
    sc.load_model(os.path.join(MESH_DIR, "cube.obj"), material((0.9, 0.9, 0.9), 0.0, 0.1),
                  (-1, 1, 0), (1, 1, 1), quad_rule)
    return sc


def frame_data(width, height, pos=(0.0, 1.0, -1.0), yaw=-90.0, pitch=0.0, rpp=4, max_bounce=6,
               scene=None):
    fd = FrameData()
    lib().orc_default_frame_constants(C.byref(fd))
    lib().orc_camera_pose(C.byref(fd), fl3(*pos), yaw, pitch, width, height)
    fd.rays_per_pixel = rpp
    fd.max_bounce = max_bounce
    if scene is not None:
        fd.sphere_count = scene.n_spheres
        fd.mesh_count = scene.n_meshes.value
    return fd


def render(fd, scene, first_frame, n_frames, accum=None, n_threads=0, want_debug=True, rows=None, profile=None):
    """Returns dict(accum, rgba8, prim_id, chain_hash, rng_state, counters).  rows=(y0, y1) renders
    only those image rows (the arrays stay full size)."""
    W, H = int(fd.window[0]), int(fd.window[1])
    y0, y1 = rows if rows is not None else (0, H)
    if accum is None:
        accum = np.zeros((H, W, 4), dtype=np.float32)
    else:
        accum = np.ascontiguousarray(accum, dtype=np.float32).copy()
    rgba8 = np.zeros((H, W, 4), dtype=np.uint8)
    prim = np.zeros((H, W), dtype=np.uint32)
    chain = np.zeros((H, W), dtype=np.uint32)
    rng = np.zeros((H, W), dtype=np.uint32)
    counters = np.zeros(N_COUNTERS, dtype=np.uint64)
    rc = lib(profile).orc_render_rows(C.byref(fd), scene.spheres, scene.tris, scene.n_tris.value, scene.meshes,
                               first_frame, n_frames,
                               accum.ctypes.data, rgba8.ctypes.data,
                               prim.ctypes.data if want_debug else None,
                               chain.ctypes.data if want_debug else None,
                               rng.ctypes.data if want_debug else None,
                               counters.ctypes.data, n_threads, y0, y1)
    if rc != 0:
        raise RuntimeError("orc_render failed: %d" % rc)
    return dict(accum=accum, rgba8=rgba8, prim_id=prim, chain_hash=chain, rng_state=rng,
                counters=dict(zip(COUNTER_NAMES, (int(c) for c in counters))))
