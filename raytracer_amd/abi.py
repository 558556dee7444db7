"""ctypes mirror of include/rtmi.h: the records with the reference's byte layouts
(RayTracingStructs.h:4-60) and the prototypes of every exported entry point.

The library is the product; if it cannot be loaded this module raises, there is no fallback."""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# RTMI_LIBRARY selects another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("RTMI_LIBRARY") or os.path.join(PKG_DIR, "librtmi.so")

N_COUNTERS = 16
COUNTER_NAMES = [
    "n_sphere_tests", "n_sphere_disc_ge0", "n_sphere_wins", "n_box_tests", "n_tri_tests",
    "n_tri_wins", "n_hit_bounces", "n_env_evals", "n_pixel_frames", "n_paths", "n_segments",
]
MISS_ID = 0xFFFFFFFF
NOALLOC_MAX_PIXELS = 1 << 21

VARIANT_MEGAKERNEL = 0
VARIANT_REGENERATE = 1
VARIANT_COMPACT = 2
VARIANT_AUTO = 3
VARIANT_STREAM = 4

ACCEL_AUTO = 0
ACCEL_NONE = 1
ACCEL_BVH = 2

PROFILE_STRICT = 0
PROFILE_FMA = 1
PROFILE_NAMES = {"strict": PROFILE_STRICT, "fma": PROFILE_FMA}

OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_NO_DEVICE = -2
ERR_HIP = -3
ERR_OUT_OF_MEMORY = -4
ERR_NO_SCENE = -5
ERR_SCENE_MISMATCH = -6
ERR_IO = -7
ERR_UNSUPPORTED = -8


class Material(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("light", C.c_float), ("smoothness", C.c_float),
                ("pad1", C.c_float), ("pad2", C.c_float), ("pad3", C.c_float)]


class Sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("material", Material)]


class Mesh(C.Structure):
    _fields_ = [("start_triangle", C.c_int32), ("num_triangles", C.c_int32),
                ("pad1", C.c_float), ("pad2", C.c_float),
                ("bounding_point1", C.c_float * 4), ("bounding_point2", C.c_float * 4),
                ("material", Material)]


class Triangle(C.Structure):
    _fields_ = [("p1", C.c_float * 4), ("p2", C.c_float * 4), ("p3", C.c_float * 4),
                ("n1", C.c_float * 4), ("n2", C.c_float * 4), ("n3", C.c_float * 4)]


class FrameData(C.Structure):
    _fields_ = [("camera_inverse_projection", C.c_float * 16),
                ("camera_inverse_view", C.c_float * 16),
                ("camera_pos", C.c_float * 4), ("camera_direction", C.c_float * 4),
                ("window", C.c_float * 2),
                ("rays_per_pixel", C.c_uint32), ("max_bounce_limit", C.c_uint32),
                ("sky_color_horizon", C.c_float * 4), ("sky_color_zenith", C.c_float * 4),
                ("sun_light_direction", C.c_float * 4), ("ground_color", C.c_float * 4),
                ("sun_focus", C.c_float), ("sun_intensity", C.c_float),
                ("frame_index", C.c_uint32), ("sphere_number", C.c_uint32),
                ("mesh_number", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("pitch", C.c_float), ("yaw", C.c_float),
                ("fov", C.c_float), ("near_plane", C.c_float), ("far_plane", C.c_float),
                ("sensitivity", C.c_float), ("moved", C.c_int),
                ("projection", C.c_float * 16), ("view", C.c_float * 16),
                ("inverse_projection", C.c_float * 16), ("inverse_view", C.c_float * 16),
                ("position", C.c_float * 3), ("forward", C.c_float * 3),
                ("up", C.c_float * 3), ("right", C.c_float * 3),
                ("last_mouse_x", C.c_double), ("last_mouse_y", C.c_double),
                ("width", C.c_uint32), ("height", C.c_uint32)]


class CameraInput(C.Structure):
    _fields_ = [("mouse_x", C.c_double), ("mouse_y", C.c_double), ("mouse_button_2", C.c_int),
                ("key_w", C.c_int), ("key_s", C.c_int), ("key_a", C.c_int), ("key_d", C.c_int),
                ("key_space", C.c_int), ("key_left_shift", C.c_int), ("key_f", C.c_int),
                ("key_c", C.c_int)]


_ctx = C.c_void_p
_f = C.POINTER(C.c_float)

# name -> (restype, argtypes); every symbol include/rtmi.h declares
PROTOTYPES = {
    "rtmi_create": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(_ctx)]),
    "rtmi_destroy": (C.c_int, [_ctx]),
    "rtmi_last_error": (C.c_char_p, [_ctx]),
    "rtmi_version": (C.c_char_p, []),
    "rtmi_upload_scene": (C.c_int, [_ctx, C.POINTER(Sphere), C.c_uint32, C.POINTER(Triangle), C.c_uint32,
                                    C.POINTER(Mesh), C.c_uint32]),
    "rtmi_render": (C.c_int, [_ctx, C.POINTER(FrameData)]),
    "rtmi_render_frames": (C.c_int, [_ctx, C.POINTER(FrameData), C.c_uint32, C.c_uint32]),
    "rtmi_render_frames_noalloc": (C.c_int, [_ctx, C.POINTER(FrameData), C.c_uint32, C.c_uint32]),
    "rtmi_set_variant": (C.c_int, [_ctx, C.c_int]),
    "rtmi_get_variant": (C.c_int, [_ctx]),
    "rtmi_set_acceleration": (C.c_int, [_ctx, C.c_int]),
    "rtmi_get_acceleration": (C.c_int, [_ctx]),
    "rtmi_bvh_node_bytes": (C.c_int, []),
    "rtmi_enable_transmission": (C.c_int, [_ctx, C.c_int]),
    "rtmi_set_grazing_cap": (C.c_int, [_ctx, C.c_float]),
    "rtmi_get_grazing_cap": (C.c_float, [_ctx]),
    "rtmi_set_profile": (C.c_int, [_ctx, C.c_int]),
    "rtmi_get_profile": (C.c_int, [_ctx]),
    "rtmi_clear_accum": (C.c_int, [_ctx]),
    "rtmi_read_accum": (C.c_int, [_ctx, C.c_void_p, C.c_size_t]),
    "rtmi_write_accum": (C.c_int, [_ctx, C.c_void_p, C.c_size_t]),
    "rtmi_read_output_rgba8": (C.c_int, [_ctx, C.c_void_p, C.c_size_t]),
    "rtmi_resolve_output": (C.c_int, [_ctx, C.c_uint32]),
    "rtmi_bind_accum": (C.c_int, [_ctx, C.c_void_p]),
    "rtmi_accum_device_ptr": (C.c_void_p, [_ctx]),
    "rtmi_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "rtmi_synchronize": (C.c_int, [_ctx]),
    "rtmi_device_count": (C.c_int, []),
    "rtmi_reduce_accum": (C.c_int, [C.POINTER(_ctx), C.c_uint32, C.c_uint32]),
    "rtmi_reduce_same_device_allowed": (C.c_int, []),
    "rtmi_shutdown": (C.c_int, []),
    "rtmi_last_render_ms": (C.c_int, [_ctx, C.POINTER(C.c_float)]),
    "rtmi_timing_reset": (C.c_int, [_ctx, C.c_int]),
    "rtmi_timing_read": (C.c_int, [_ctx, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "rtmi_enable_validation": (C.c_int, [_ctx, C.c_int]),
    "rtmi_read_hit_ids": (C.c_int, [_ctx, C.c_void_p, C.c_size_t]),
    "rtmi_read_chain_hash": (C.c_int, [_ctx, C.c_void_p, C.c_size_t]),
    "rtmi_read_rng_state": (C.c_int, [_ctx, C.c_void_p, C.c_size_t]),
    "rtmi_get_counters": (C.c_int, [_ctx, C.c_void_p, C.c_size_t]),
    "rtmi_reset_counters": (C.c_int, [_ctx]),
    "rtmi_selftest_math": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtmi_selftest_env_light": (C.c_int, [_ctx, C.POINTER(FrameData), C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "rtmi_selftest_exhaustive": (C.c_int, [_ctx, C.c_int, C.c_float, C.c_void_p]),
    "rtmi_assume_compute_units": (C.c_int, [_ctx, C.c_int]),
    "rtmi_paths_pixels_per_wave_for": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "rtmi_paths_pixels_per_wave": (C.c_int, [_ctx, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "rtmi_selftest_walk": (C.c_int, [_ctx, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p,
                                     C.c_void_p, C.POINTER(C.c_float)]),
    "rtmi_selftest_closest_hit": (C.c_int, [_ctx, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_void_p]),
    "rtmi_load_model": (C.c_int, [C.c_char_p, C.POINTER(Triangle), C.POINTER(C.c_uint32), C.c_uint32,
                                  C.POINTER(Mesh), C.POINTER(C.c_uint32), C.c_uint32,
                                  C.POINTER(Material), _f, _f, C.c_int]),
    "rtmi_checkpoint_save": (C.c_int, [_ctx, C.c_char_p, C.c_uint32]),
    "rtmi_checkpoint_load": (C.c_int, [_ctx, C.c_char_p, C.POINTER(C.c_uint32)]),
    "rtmi_scene_save": (C.c_int, [C.c_char_p, C.POINTER(Sphere), C.c_uint32, C.POINTER(Triangle), C.c_uint32,
                                  C.POINTER(Mesh), C.c_uint32]),
    "rtmi_scene_load": (C.c_int, [C.c_char_p, C.POINTER(Sphere), C.POINTER(C.c_uint32), C.c_uint32,
                                  C.POINTER(Triangle), C.POINTER(C.c_uint32), C.c_uint32,
                                  C.POINTER(Mesh), C.POINTER(C.c_uint32), C.c_uint32]),
    "rtmi_bvh_build": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p,
                                 C.POINTER(C.c_uint32)]),
    "rtmi_bvh_quantize": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rtmi_default_frame_data": (None, [C.POINTER(FrameData)]),
    "rtmi_camera_init": (None, [C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_double, C.c_double]),
    "rtmi_camera_set_pose": (None, [C.POINTER(Camera), _f, C.c_float, C.c_float]),
    "rtmi_camera_update": (None, [C.POINTER(Camera), C.c_double, C.POINTER(CameraInput)]),
    "rtmi_frame_advance": (None, [C.POINTER(FrameData), C.POINTER(Camera), C.c_int]),
    "rtmi_write_ppm": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]),
    "rtmi_write_pfm": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rtmi_write_exr": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
}

_lib = None


def load():
    """Load librtmi.so (built in-tree by raytracer_amd/csrc/Makefile) and bind every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "raytracer_amd: %s is missing; build it with `make -C raytracer_amd/csrc` "
            "(there is no fallback path)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    # RTMI_HOST_ONLY=1: a sanitizer build of the host side alone (raytracer_amd/csrc/Makefile, host-asan /
    # host-tsan) exports no device entry points; only what it has is bound
    host_only = os.environ.get("RTMI_HOST_ONLY") == "1"
    for name, (res, args) in PROTOTYPES.items():
        if host_only and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
