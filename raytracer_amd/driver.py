"""The reference's frame loop (main.cpp:236-444) with scripted input in place of the GLFW window:
per frame, CameraFPS::Update (CameraFPS.cpp:33-88), the frame-index policy (main.cpp:332-335,
361-363: reset key -> 0, +1 every frame, camera moved -> 1) and one dispatch."""
import ctypes as C

from . import abi


def camera_input(keys="", mouse=(0.0, 0.0), button2=False):
    """keys: any of W S A D F C, '^' for space, 'v' for left shift."""
    inp = abi.CameraInput()
    inp.mouse_x, inp.mouse_y = mouse
    inp.mouse_button_2 = 1 if button2 else 0
    names = {"W": "key_w", "S": "key_s", "A": "key_a", "D": "key_d", "F": "key_f", "C": "key_c",
             "^": "key_space", "v": "key_left_shift"}
    for k in keys:
        setattr(inp, names[k], 1)
    return inp


def replay(renderer, fd, cam, steps):
    """steps: iterable of (frames, dt, CameraInput, reset_key).  Renders every frame with
    rtmi_render and returns, per frame, (frame_index, (x, y, z), yaw, pitch) as used for it."""
    lib = abi.load()
    log = []
    for frames, dt, inp, reset in steps:
        for _ in range(frames):
            lib.rtmi_camera_update(C.byref(cam), dt, C.byref(inp))
            lib.rtmi_frame_advance(C.byref(fd), C.byref(cam), 1 if reset else 0)
            renderer.render(fd)
            log.append((fd.frame_index, tuple(cam.position), cam.yaw, cam.pitch))
    return log
