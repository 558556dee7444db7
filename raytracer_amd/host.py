"""Python host layer over the C ABI: scene building, camera/frame-data helpers and a Renderer
object.  It mirrors what the reference's main.cpp does around the dispatch (scene literals
main.cpp:161-185, constants main.cpp:116-153, per-frame policy main.cpp:354-363) and contains no
rendering arithmetic of its own: every pixel comes from librtmi.so."""
import ctypes as C
import os

import numpy as np

from . import abi

MESH_DIR = os.path.join(os.path.dirname(abi.PKG_DIR), "assets", "meshes")

# (position, yaw degrees, pitch degrees)
DEFAULT_POSE = ((0.0, 1.0, -1.0), -90.0, 0.0)   # CameraFPS.cpp:13, CameraFPS.h:15-16; sees no geometry
BENCH_POSE = ((0.0, 1.5, -4.5), 90.0, -15.0)    # looks at the scene from in front and above


# numpy views of the reference's records (RayTracingStructs.h:4-60): 32, 48, 96 and 80 bytes
MATERIAL_DTYPE = np.dtype([("color", "<f4", 3), ("light", "<f4"), ("smoothness", "<f4"), ("pad", "<f4", 3)])
SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("material", MATERIAL_DTYPE)])
TRIANGLE_DTYPE = np.dtype([("p1", "<f4", 4), ("p2", "<f4", 4), ("p3", "<f4", 4),
                           ("n1", "<f4", 4), ("n2", "<f4", 4), ("n3", "<f4", 4)])
MESH_DTYPE = np.dtype([("start_triangle", "<i4"), ("num_triangles", "<i4"), ("pad", "<f4", 2),
                       ("bounding_point1", "<f4", 4), ("bounding_point2", "<f4", 4),
                       ("material", MATERIAL_DTYPE)])
assert (MATERIAL_DTYPE.itemsize, SPHERE_DTYPE.itemsize, TRIANGLE_DTYPE.itemsize, MESH_DTYPE.itemsize) == (32, 48, 96, 80)


class RtmiError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rtmi error %d: %s" % (code, message))
        self.code = code


def material(color, light=0.0, smoothness=0.0):
    m = abi.Material()
    m.color[:] = color
    m.light = light
    m.smoothness = smoothness
    return m


def sphere(center, radius, mat):
    s = abi.Sphere()
    s.center[:] = center
    s.radius = radius
    s.material = mat
    return s


class Scene:
    """Sphere[] / Triangle[] / Mesh[] in the reference's AoS records."""

    def __init__(self, spheres=(), triangle_capacity=1 << 16, mesh_capacity=1 << 10):
        self.spheres = list(spheres)
        self._sphere_arr = None  # set by from_records: a ready ctypes array instead of the list
        self._tris = (abi.Triangle * triangle_capacity)()
        self._meshes = (abi.Mesh * mesh_capacity)()
        self._n_tris = C.c_uint32(0)
        self._n_meshes = C.c_uint32(0)

    @classmethod
    def from_records(cls, spheres, triangles, meshes):
        """Scene over numpy record arrays with the dtypes SPHERE_DTYPE / TRIANGLE_DTYPE / MESH_DTYPE
        (the byte layouts of RayTracingStructs.h:4-60); for scenes too large for Python lists."""
        spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
        triangles = np.ascontiguousarray(triangles, dtype=TRIANGLE_DTYPE)
        meshes = np.ascontiguousarray(meshes, dtype=MESH_DTYPE)
        sc = cls((), triangle_capacity=max(1, len(triangles)), mesh_capacity=max(1, len(meshes)))
        sc._sphere_arr = (abi.Sphere * max(1, len(spheres)))()
        sc._n_sphere_arr = len(spheres)
        C.memmove(sc._sphere_arr, spheres.ctypes.data, spheres.nbytes)
        C.memmove(sc._tris, triangles.ctypes.data, triangles.nbytes)
        C.memmove(sc._meshes, meshes.ctypes.data, meshes.nbytes)
        sc._n_tris.value = len(triangles)
        sc._n_meshes.value = len(meshes)
        return sc

    @property
    def n_spheres(self):
        return self._n_sphere_arr if self._sphere_arr is not None else len(self.spheres)

    @property
    def n_triangles(self):
        return self._n_tris.value

    @property
    def n_meshes(self):
        return self._n_meshes.value

    @property
    def triangles(self):
        return self._tris

    @property
    def meshes(self):
        return self._meshes

    def sphere_array(self):
        if self._sphere_arr is not None:
            return self._sphere_arr
        return (abi.Sphere * max(1, len(self.spheres)))(*self.spheres)

    def save(self, path):
        """Write the scene as a text scene file (rtmi_scene_save)."""
        rc = abi.load().rtmi_scene_save(os.fspath(path).encode(), self.sphere_array(), self.n_spheres,
                                        self._tris, self.n_triangles, self._meshes, self.n_meshes)
        if rc < 0:
            raise RtmiError(rc, "rtmi_scene_save(%s)" % path)

    @classmethod
    def load(cls, path):
        """Read a text scene file (rtmi_scene_load)."""
        lib = abi.load()
        ns, nt, nm = C.c_uint32(), C.c_uint32(), C.c_uint32()
        p = os.fspath(path).encode()
        rc = lib.rtmi_scene_load(p, None, C.byref(ns), 0, None, C.byref(nt), 0, None, C.byref(nm), 0)
        if rc < 0:
            raise RtmiError(rc, "rtmi_scene_load(%s)" % path)
        sc = cls((), triangle_capacity=max(1, nt.value), mesh_capacity=max(1, nm.value))
        sc._sphere_arr = (abi.Sphere * max(1, ns.value))()
        rc = lib.rtmi_scene_load(p, sc._sphere_arr, C.byref(ns), max(1, ns.value), sc._tris, C.byref(nt),
                                 max(1, nt.value), sc._meshes, C.byref(nm), max(1, nm.value))
        if rc < 0:
            raise RtmiError(rc, "rtmi_scene_load(%s)" % path)
        sc._n_sphere_arr = ns.value
        sc._n_tris.value = nt.value
        sc._n_meshes.value = nm.value
        return sc

    def load_model(self, path, mat, translate=(0.0, 0.0, 0.0), scale=(1.0, 1.0, 1.0), quad_rule=0):
        """LoadModel (ModelLoader.h:6-11).  Returns the number of triangles appended."""
        rc = abi.load().rtmi_load_model(
            os.fspath(path).encode(), self._tris, C.byref(self._n_tris), len(self._tris),
            self._meshes, C.byref(self._n_meshes), len(self._meshes), C.byref(mat),
            (C.c_float * 3)(*translate), (C.c_float * 3)(*scale), quad_rule)
        if rc < 0:
            raise RtmiError(rc, "rtmi_load_model(%s)" % path)
        return rc


def reference_scene(mesh_dir=MESH_DIR, quad_rule=0):
    """main.cpp:161-185: one sphere, plane.obj scaled (2,1,2), cube.obj moved to (-1,1,0)."""
    sc = Scene([sphere((1.0, 1.0, 0.0), 0.5, material((1.0, 1.0, 1.0), 0.0, 0.1))])
    sc.load_model(os.path.join(mesh_dir, "plane.obj"), material((1.0, 1.0, 1.0), 0.0, 0.8),
                  (0.0, 0.0, 0.0), (2.0, 1.0, 2.0), quad_rule)
    sc.load_model(os.path.join(mesh_dir, "cube.obj"), material((0.9, 0.9, 0.9), 0.0, 0.1),
                  (-1.0, 1.0, 0.0), (1.0, 1.0, 1.0), quad_rule)
    return sc


# Spheres-only benchmark scene: the reference's sphere, a ground sphere, and a 3x3 grid whose
# smoothness runs 0..1, two of them emitting.  A fixed literal table, no random numbers.
_SPHERE_SCENE_TABLE = [
    # center,               radius,  color,              light, smoothness
    ((1.0, 1.0, 0.0),        0.5,    (1.0, 1.0, 1.0),    0.0,   0.1),
    ((0.0, -1000.0, 0.0),    1000.0, (0.8, 0.8, 0.8),    0.0,   0.0),
    ((-2.4, 0.5, -1.2),      0.5,    (0.9, 0.2, 0.2),    0.0,   0.0),
    ((-1.2, 0.5, -1.2),      0.5,    (0.2, 0.9, 0.2),    0.0,   0.125),
    ((0.0, 0.5, -1.2),       0.5,    (1.0, 0.9, 0.7),    2.0,   0.25),
    ((-2.4, 0.5, 0.0),       0.5,    (0.2, 0.2, 0.9),    0.0,   0.375),
    ((-1.2, 0.5, 0.0),       0.5,    (0.9, 0.9, 0.2),    0.0,   0.5),
    ((0.0, 0.5, 0.0),        0.5,    (0.9, 0.2, 0.9),    0.0,   0.625),
    ((-2.4, 0.5, 1.2),       0.5,    (0.7, 0.9, 1.0),    2.0,   0.75),
    ((-1.2, 0.5, 1.2),       0.5,    (0.2, 0.9, 0.9),    0.0,   0.875),
    ((0.0, 0.5, 1.2),        0.5,    (0.9, 0.9, 0.9),    0.0,   1.0),
]


def sphere_scene():
    return Scene([sphere(c, r, material(col, light, sm)) for c, r, col, light, sm in _SPHERE_SCENE_TABLE])


def make_camera(width, height, pose=DEFAULT_POSE):
    cam = abi.Camera()
    lib = abi.load()
    lib.rtmi_camera_init(C.byref(cam), width, height, 0.0, 0.0)
    if pose is not None and pose != DEFAULT_POSE:
        pos, yaw, pitch = pose
        lib.rtmi_camera_set_pose(C.byref(cam), (C.c_float * 3)(*pos), yaw, pitch)
        cam.moved = 0
    return cam


def make_frame_data(width, height, scene=None, pose=DEFAULT_POSE, rays_per_pixel=4, max_bounce=6):
    """Frame constants of main.cpp:116-153 plus the camera block of main.cpp:355-360.
    frame_index is left at 0 (main.cpp:120); the frame loop or render_frames supplies it."""
    lib = abi.load()
    fd = abi.FrameData()
    lib.rtmi_default_frame_data(C.byref(fd))
    cam = make_camera(width, height, pose)
    lib.rtmi_frame_advance(C.byref(fd), C.byref(cam), 0)
    fd.frame_index = 0
    fd.rays_per_pixel = rays_per_pixel
    fd.max_bounce_limit = max_bounce
    if scene is not None:
        fd.sphere_number = scene.n_spheres
        fd.mesh_number = scene.n_meshes
    return fd


class Renderer:
    """One context = one GPU, one image extent (rtmi_create ... rtmi_destroy)."""

    def __init__(self, width, height, device=0, variant=abi.VARIANT_AUTO, profile=None):
        """profile: "strict" (default) or "fma", the arithmetic profile scenes are uploaded and rendered
        under (rtmi_set_profile); None takes it from the environment variable RTMI_PROFILE."""
        self._lib = abi.load()
        self._ctx = C.c_void_p()
        self.width, self.height = int(width), int(height)
        rc = self._lib.rtmi_create(device, self.width, self.height, C.byref(self._ctx))
        if rc != 0:
            msg = self._lib.rtmi_last_error(None)
            self._ctx = None
            raise RtmiError(rc, msg.decode() if msg else "rtmi_create")
        self.set_variant(variant)
        if profile is None:
            profile = os.environ.get("RTMI_PROFILE") or "strict"
        self.set_profile(profile)

    def _check(self, rc):
        if rc < 0:
            msg = self._lib.rtmi_last_error(self._ctx)
            raise RtmiError(rc, msg.decode() if msg else "")
        return rc

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.rtmi_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_variant(self, variant):
        self._check(self._lib.rtmi_set_variant(self._ctx, variant))

    def set_profile(self, profile):
        """"strict" / "fma" (or abi.PROFILE_*) for the next upload_scene."""
        if isinstance(profile, str):
            if profile not in abi.PROFILE_NAMES:
                raise ValueError("unknown arithmetic profile %r" % (profile,))
            profile = abi.PROFILE_NAMES[profile]
        self._check(self._lib.rtmi_set_profile(self._ctx, profile))

    def profile(self):
        """Name of the profile the uploaded scene is rendered under (before an upload: the one set)."""
        v = self._check(self._lib.rtmi_get_profile(self._ctx))
        return "fma" if v == abi.PROFILE_FMA else "strict"

    def set_acceleration(self, mode):
        """abi.ACCEL_AUTO / ACCEL_NONE / ACCEL_BVH for the next upload_scene."""
        self._check(self._lib.rtmi_set_acceleration(self._ctx, mode))

    def enable_transmission(self, on=True):
        """The transmitting-material extension (rtmi_enable_transmission) for the next upload_scene: materials
        whose pad[0] is above 0.5 refract, pad[1] is the index of refraction."""
        self._check(self._lib.rtmi_enable_transmission(self._ctx, 1 if on else 0))

    def set_grazing_cap(self, cap):
        """Largest extra margin (fraction of the distance) the boxes of a mesh of small triangles are given for
        a ray that runs almost inside a triangle's plane (rtmi_set_grazing_cap); 0 = none, 1.0 = the proven
        bound, whatever it costs.  A kernel argument: applies from the next launch on."""
        self._check(self._lib.rtmi_set_grazing_cap(self._ctx, float(cap)))

    def grazing_cap(self):
        return float(self._lib.rtmi_get_grazing_cap(self._ctx))

    def acceleration(self):
        """What the uploaded scene uses: abi.ACCEL_NONE or abi.ACCEL_BVH."""
        return self._check(self._lib.rtmi_get_acceleration(self._ctx))

    def upload_scene(self, scene):
        self._check(self._lib.rtmi_upload_scene(
            self._ctx, scene.sphere_array(), scene.n_spheres, scene.triangles, scene.n_triangles,
            scene.meshes, scene.n_meshes))

    def render(self, fd):
        self._check(self._lib.rtmi_render(self._ctx, C.byref(fd)))

    def render_frames(self, fd, first_frame, n_frames):
        self._check(self._lib.rtmi_render_frames(self._ctx, C.byref(fd), first_frame, n_frames))

    def render_frames_noalloc(self, fd, first_frame, n_frames):
        """rtmi_render_frames_noalloc: the same images from a call that allocates nothing."""
        self._check(self._lib.rtmi_render_frames_noalloc(self._ctx, C.byref(fd), first_frame, n_frames))

    def synchronize(self):
        self._check(self._lib.rtmi_synchronize(self._ctx))

    def last_render_ms(self):
        ms = C.c_float()
        self._check(self._lib.rtmi_last_render_ms(self._ctx, C.byref(ms)))
        return ms.value

    def timing_reset(self, enable=True):
        self._check(self._lib.rtmi_timing_reset(self._ctx, 1 if enable else 0))

    def timing_read(self):
        """(summed device milliseconds, number of render calls) since timing_reset."""
        total, n = C.c_double(), C.c_uint32()
        self._check(self._lib.rtmi_timing_read(self._ctx, C.byref(total), C.byref(n)))
        return total.value, n.value

    def clear_accum(self):
        self._check(self._lib.rtmi_clear_accum(self._ctx))

    def save_checkpoint(self, path, frames_done):
        self._check(self._lib.rtmi_checkpoint_save(self._ctx, os.fspath(path).encode(), frames_done))

    def load_checkpoint(self, path):
        """Returns the number of frames the snapshot holds; continue with frame_index = that + 1."""
        n = C.c_uint32()
        self._check(self._lib.rtmi_checkpoint_load(self._ctx, os.fspath(path).encode(), C.byref(n)))
        return n.value

    def bind_accum(self, device_ptr):
        self._check(self._lib.rtmi_bind_accum(self._ctx, device_ptr))

    def set_stream(self, stream_handle):
        self._check(self._lib.rtmi_set_stream(self._ctx, stream_handle))

    def resolve_output(self, divisor):
        self._check(self._lib.rtmi_resolve_output(self._ctx, divisor))

    def enable_validation(self, on=True):
        self._check(self._lib.rtmi_enable_validation(self._ctx, 1 if on else 0))

    def reset_counters(self):
        self._check(self._lib.rtmi_reset_counters(self._ctx))

    def _read(self, fn, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        self._check(fn(self._ctx, out.ctypes.data, out.nbytes))
        return out

    def read_accum(self):
        return self._read(self._lib.rtmi_read_accum, np.float32, (self.height, self.width, 4))

    def write_accum(self, accum):
        a = np.ascontiguousarray(accum, dtype=np.float32)
        assert a.shape == (self.height, self.width, 4)
        self._check(self._lib.rtmi_write_accum(self._ctx, a.ctypes.data, a.nbytes))

    def read_output_rgba8(self):
        return self._read(self._lib.rtmi_read_output_rgba8, np.uint8, (self.height, self.width, 4))

    def read_hit_ids(self):
        return self._read(self._lib.rtmi_read_hit_ids, np.uint32, (self.height, self.width))

    def read_chain_hash(self):
        return self._read(self._lib.rtmi_read_chain_hash, np.uint32, (self.height, self.width))

    def read_rng_state(self):
        return self._read(self._lib.rtmi_read_rng_state, np.uint32, (self.height, self.width))

    def counters(self):
        out = np.zeros(abi.N_COUNTERS, dtype=np.uint64)
        self._check(self._lib.rtmi_get_counters(self._ctx, out.ctypes.data, abi.N_COUNTERS))
        return dict(zip(abi.COUNTER_NAMES, (int(v) for v in out)))

    def selftest_closest_hit(self, sphere_number, mesh_number, origins, directions):
        """(t, prim_id) of each ray against the uploaded scene; rays are [n, 3] arrays."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        assert o.shape == d.shape
        t = np.empty(len(o), dtype=np.float32)
        prim = np.empty(len(o), dtype=np.uint32)
        self._check(self._lib.rtmi_selftest_closest_hit(self._ctx, sphere_number, mesh_number, o.ctypes.data,
                                                        d.ctypes.data, len(o), t.ctypes.data, prim.ctypes.data))
        return t, prim

    def assume_compute_units(self, n):
        """Scheduling decisions take the device to have n compute units (0 = what it reports); results never change."""
        self._check(self._lib.rtmi_assume_compute_units(self._ctx, int(n)))

    def paths_pixels_per_wave(self, n_listed, n_frames, rays_per_pixel=1, mesh_number=0):
        """(pixels per wave, neighbours?) the compacting pair takes for a work list of n_listed pixels (rtmi_paths_pixels_per_wave)."""
        pixels, together = C.c_uint32(0), C.c_int(0)
        self._check(self._lib.rtmi_paths_pixels_per_wave(self._ctx, n_listed, n_frames, rays_per_pixel, mesh_number,
                                                         C.byref(pixels), C.byref(together)))
        return int(pixels.value), bool(together.value)

    def selftest_walk(self, sphere_number, mesh_number, origins, directions, repeats=1):
        """(t, prim_id, ms): the rays, in the order given, through the walk-only streaming kernel (hierarchy scenes)."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        assert o.shape == d.shape
        t = np.empty(len(o), dtype=np.float32)
        prim = np.empty(len(o), dtype=np.uint32)
        ms = C.c_float(0.0)
        self._check(self._lib.rtmi_selftest_walk(self._ctx, sphere_number, mesh_number, o.ctypes.data, d.ctypes.data, len(o),
                                                 repeats, t.ctypes.data, prim.ctypes.data, C.byref(ms)))
        return t, prim, float(ms.value)

    def selftest_env_light(self, fd, directions, wave_form):
        """Sky radiance [n, 3] of directions [n, 3] under fd's sky constants; wave_form picks the routine."""
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        out = np.empty_like(d)
        self._check(self._lib.rtmi_selftest_env_light(self._ctx, C.byref(fd), d.ctypes.data, len(d), 1 if wave_form else 0,
                                                      out.ctypes.data))
        return out

    def selftest_exhaustive(self, op, divisor=1.0):
        """(patterns tested, mismatches, smallest mismatching bit pattern) of a short exact routine
        against the general operation: op 0 sqrt, 1 reciprocal, 2 division by `divisor`."""
        out = np.zeros(3, dtype=np.uint64)
        self._check(self._lib.rtmi_selftest_exhaustive(self._ctx, op, divisor, out.ctypes.data))
        return int(out[0]), int(out[1]), int(out[2])

    def selftest_math(self, op, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        yp = None
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.float32)
            assert y.shape == x.shape
            yp = y.ctypes.data
        self._check(self._lib.rtmi_selftest_math(self._ctx, op, x.ctypes.data, yp, out.ctypes.data, x.size))
        return out


# per-event FLOP costs of the algorithm (one per add/sub/mul/min/max/div/sqrt/convert)
FLOP_COSTS = {
    "n_sphere_tests": 25, "n_sphere_disc_ge0": 5, "n_sphere_wins": 18, "n_box_tests": 25,
    "n_tri_tests": 55, "n_tri_wins": 30, "n_hit_bounces": 58, "n_env_evals": 63,
    "n_pixel_frames": 74 + 9,
}


def algorithmic_flops(counters):
    return sum(cost * counters[name] for name, cost in FLOP_COSTS.items())
