"""ctypes access to tests/emu/libemu*.so: the library's own hierarchy traversal (rt_bvh_device.hpp) and scene
packing (host/rt_pack.cpp) compiled for the host, one ray at a time.  Test infrastructure."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "emu")
_libs = {}


def build():
    subprocess.check_call(["make", "-C", EMU_DIR], stdout=subprocess.DEVNULL)


def lib(kind="strict"):
    """kind: "strict", "fma" or "asan" (the last only inside a process started with the sanitizer runtime preloaded)."""
    if kind not in _libs:
        build()
        name = {"strict": "libemu.so", "fma": "libemu_fma.so", "asan": "libemu_asan.so"}[kind]
        L = C.CDLL(os.path.join(EMU_DIR, name))
        L.emu_closest_hit.restype = C.c_int
        L.emu_closest_hit.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_float,
                                      C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_char_p]
        _libs[kind] = L
    return _libs[kind]


def closest_hit(scene, origins, directions, use_bvh, cap=2.0 ** -9, kind="strict"):
    """(t, prim) per ray of a raytracer_amd.Scene: through the hierarchy (use_bvh) or the exhaustive loops."""
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
    t = np.empty(len(o), dtype=np.float32)
    prim = np.empty(len(o), dtype=np.uint32)
    msg = C.create_string_buffer(256)
    rc = lib(kind).emu_closest_hit(C.cast(scene.sphere_array(), C.c_void_p), scene.n_spheres, C.cast(scene.triangles, C.c_void_p),
                                   scene.n_triangles, C.cast(scene.meshes, C.c_void_p), scene.n_meshes, 1 if use_bvh else 0, cap,
                                   o.ctypes.data, d.ctypes.data, len(o), t.ctypes.data, prim.ctypes.data, msg)
    if rc != 0:
        raise RuntimeError("emu_closest_hit: %d %s" % (rc, msg.value.decode()))
    return t, prim
