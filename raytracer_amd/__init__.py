"""raytracer_amd — MI355X (gfx950) implementation of VoidRune/Raytracer's path-tracing dispatch.

The product is librtmi.so (HIP kernels + C ABI, include/rtmi.h).  This package binds it with
ctypes; importing `abi.load()` fails loudly when the library has not been built."""
from . import abi  # noqa: F401
from . import driver  # noqa: F401
from . import scenes  # noqa: F401
from .host import (BENCH_POSE, DEFAULT_POSE, Renderer, RtmiError, Scene, algorithmic_flops,  # noqa: F401
                   make_camera, make_frame_data, material, reference_scene, sphere, sphere_scene)

__all__ = ["abi", "driver", "scenes", "Renderer", "RtmiError", "Scene", "make_camera", "make_frame_data", "material", "sphere",
           "reference_scene", "sphere_scene", "algorithmic_flops", "DEFAULT_POSE", "BENCH_POSE"]
