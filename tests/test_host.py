"""Host side kept from the reference, exercised through the C ABI on the CPU: LoadModel, camera,
frame policy, image writers.  The oracle's independent C restatement is the comparison."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import cases


def test_reference_scene_matches_oracle(rtmi, oracle):
    for rule in (0, 1):
        a = rtmi.reference_scene(quad_rule=rule)
        b = oracle.reference_scene(quad_rule=rule)
        assert (a.n_spheres, a.n_triangles, a.n_meshes) == (1, 14, 2)
        assert bytes(a.triangles)[:96 * 14] == bytes(b.tris)[:96 * 14]
        assert bytes(a.meshes)[:80 * 2] == bytes(b.meshes)[:80 * 2]
        assert bytes(a.sphere_array())[:48] == bytes(b.spheres)[:48]


def test_load_model_missing_file_appends_nothing(rtmi):
    sc = rtmi.Scene()
    with pytest.raises(rtmi.RtmiError) as e:
        sc.load_model("/nonexistent/none.obj", rtmi.material((1, 1, 1)))
    assert e.value.code == rtmi.abi.ERR_IO
    assert sc.n_triangles == 0 and sc.n_meshes == 0


def test_load_model_formats(rtmi, tmp_path):
    p = tmp_path / "m.obj"
    p.write_text("# comment\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 2 0\nvn 0 0 1\n"
                 "f 1 2 3\n"            # plain indices, no normals
                 "f 1/1 2/1 3/1\n"      # v/vt
                 "f 1//1 2//1 3//1\n"   # v//vn
                 "f -5//-1 -4//-1 -3//-1\n"  # negative (relative) indices
                 "f 1 2 3 4 5\n")       # pentagon -> fan of 3
    sc = rtmi.Scene()
    n = sc.load_model(p, rtmi.material((1, 1, 1)), translate=(10, 0, 0), scale=(2, 3, 4))
    assert n == 7 and sc.n_meshes == 1
    t = sc.triangles
    assert list(t[0].p2)[:3] == [12.0, 0.0, 0.0] and list(t[0].n1)[:3] == [0.0, 0.0, 0.0]
    assert list(t[2].n1)[:3] == [0.0, 0.0, 1.0]
    assert bytes(t[3])[:96] == bytes(t[2])[:96]
    assert list(t[6].p3)[:3] == [11.0, 6.0, 0.0]
    m = sc.meshes[0]
    assert list(m.bounding_point1)[:3] == [10.0, 0.0, 0.0] and list(m.bounding_point2)[:3] == [12.0, 6.0, 0.0]
    # an empty file still appends a mesh record with the seed box, as the reference does
    e = tmp_path / "e.obj"
    e.write_text("# nothing\n")
    assert sc.load_model(e, rtmi.material((1, 1, 1))) == 0
    m = sc.meshes[1]
    assert (m.start_triangle, m.num_triangles) == (7, 0)
    assert list(m.bounding_point1)[:3] == [1024.0] * 3 and list(m.bounding_point2)[:3] == [-1024.0] * 3


def test_load_model_capacity(rtmi):
    sc = rtmi.Scene(triangle_capacity=5)
    with pytest.raises(rtmi.RtmiError) as e:
        sc.load_model(os.path.join(rtmi.host.MESH_DIR, "cube.obj"), rtmi.material((1, 1, 1)))
    assert e.value.code == rtmi.abi.ERR_OUT_OF_MEMORY and sc.n_triangles == 0


@pytest.mark.parametrize("pose", [cases.DEFAULT_POSE, cases.BENCH_POSE, ((3.0, 2.0, 1.0), 37.5, 20.0)])
def test_frame_data_matches_oracle(rtmi, oracle, pose):
    sc = rtmi.reference_scene()
    osc = oracle.reference_scene()
    for w, h in ((1280, 720), (1920, 1080), (37, 19)):
        fd = rtmi.make_frame_data(w, h, sc, pose, 4, 6)
        ofd = oracle.frame_data(w, h, *pose, rpp=4, max_bounce=6, scene=osc)
        assert bytes(fd) == bytes(ofd)


def test_default_frame_constants(rtmi):
    fd = rtmi.abi.FrameData()
    rtmi.abi.load().rtmi_default_frame_data(C.byref(fd))
    assert (fd.rays_per_pixel, fd.max_bounce_limit, fd.frame_index) == (4, 6, 0)
    assert np.allclose(fd.sun_light_direction[:], [-0.4, -0.4, -0.4, 0.0])
    assert np.allclose(fd.sky_color_horizon[:], [0.7, 0.3, 0.1, 0.0])
    assert np.allclose(fd.sky_color_zenith[:], [0.2, 0.56, 0.95, 0.0])
    assert np.allclose(fd.ground_color[:], [0.9, 0.9, 0.9, 0.0])
    assert fd.sun_focus == 1.0 and fd.sun_intensity == 1.0


def _update(rtmi, cam, dt=0.5, **keys):
    inp = rtmi.abi.CameraInput()
    inp.mouse_x, inp.mouse_y = cam.last_mouse_x, cam.last_mouse_y
    for k, v in keys.items():
        setattr(inp, k, v)
    rtmi.abi.load().rtmi_camera_update(C.byref(cam), dt, C.byref(inp))


def test_camera_defaults_and_motion(rtmi):
    cam = rtmi.make_camera(1280, 720)
    assert list(cam.position) == [0.0, 1.0, -1.0] and cam.yaw == -90.0 and cam.pitch == 0.0 and cam.moved == 0
    assert np.allclose(cam.forward[:], [0.0, 0.0, -1.0], atol=1e-6)
    assert np.allclose(cam.right[:], [1.0, 0.0, 0.0], atol=1e-6)   # cross(forward, up)
    _update(rtmi, cam)
    assert cam.moved == 0 and list(cam.position) == [0.0, 1.0, -1.0]
    # W moves along -forward (CameraFPS.cpp:62), speed 2, dt 0.5 -> +1 in z
    _update(rtmi, cam, key_w=1)
    assert cam.moved == 1 and np.allclose(cam.position[:], [0.0, 1.0, 0.0], atol=1e-6)
    _update(rtmi, cam, key_s=1, key_f=1)                            # speed 10
    assert np.allclose(cam.position[:], [0.0, 1.0, -5.0], atol=1e-5)
    _update(rtmi, cam, key_a=1, key_c=1)                            # A = +right, speed 0.5
    assert np.allclose(cam.position[:], [0.25, 1.0, -5.0], atol=1e-5)
    _update(rtmi, cam, key_d=1, key_space=1)                        # D = -right, space = +y (not normalised together)
    assert np.allclose(cam.position[:], [-0.75, 2.0, -5.0], atol=1e-5)
    _update(rtmi, cam, key_left_shift=1)
    assert np.allclose(cam.position[:], [-0.75, 1.0, -5.0], atol=1e-5)
    # the view follows the position
    iv = np.array(cam.inverse_view[:]).reshape(4, 4).T
    assert np.allclose(iv[:3, 3], cam.position[:], atol=1e-5)
    _update(rtmi, cam)
    assert cam.moved == 0


def test_camera_mouse_look(rtmi):
    cam = rtmi.make_camera(640, 360)
    inp = rtmi.abi.CameraInput()
    inp.mouse_x, inp.mouse_y, inp.mouse_button_2 = 50.0, 1000.0, 0
    rtmi.abi.load().rtmi_camera_update(C.byref(cam), 0.1, C.byref(inp))
    assert cam.moved == 0 and cam.yaw == -90.0            # button up: only the cursor is remembered
    inp.mouse_x, inp.mouse_y, inp.mouse_button_2 = 150.0, 2000.0, 1
    rtmi.abi.load().rtmi_camera_update(C.byref(cam), 0.1, C.byref(inp))
    assert cam.moved == 1 and abs(cam.yaw - (-70.0)) < 1e-4 and cam.pitch == 89.0   # +100 px * 0.2, pitch clamped
    f = [math.cos(math.radians(-70.0)) * math.cos(math.radians(89.0)), math.sin(math.radians(89.0)),
         math.sin(math.radians(-70.0)) * math.cos(math.radians(89.0))]
    assert np.allclose(cam.forward[:], f, atol=1e-5)
    inp.mouse_x = 150.0 + 5 * 360 * 5                     # yaw wraps with fmod
    rtmi.abi.load().rtmi_camera_update(C.byref(cam), 0.1, C.byref(inp))
    assert abs(cam.yaw - 290.0) < 1e-2                     # fmod(-70 + 1800, 360), same direction as -70
    assert np.allclose(cam.forward[:], f, atol=1e-4)


def test_frame_policy(rtmi):
    lib = rtmi.abi.load()
    cam = rtmi.make_camera(320, 180)
    fd = rtmi.abi.FrameData()
    lib.rtmi_default_frame_data(C.byref(fd))
    seen = []
    for step in range(8):
        _update(rtmi, cam, key_w=1 if step == 4 else 0)
        lib.rtmi_frame_advance(C.byref(fd), C.byref(cam), 1 if step == 6 else 0)
        seen.append(fd.frame_index)
    # ++ each frame, 1 when the camera moved, Q zeroes it so the next value is 1
    assert seen == [1, 2, 3, 4, 1, 2, 1, 2]
    assert list(fd.window) == [320.0, 180.0] and fd.camera_pos[3] == 0.0


def test_ppm_and_pfm_writers(rtmi, tmp_path):
    lib = rtmi.abi.load()
    w, h = 3, 2
    img = np.zeros((h, w, 4), dtype=np.uint8)
    img[0, :, 0] = [10, 20, 30]       # buffer row 0 = bottom of the view
    img[1, :, 1] = [100, 150, 200]
    p = tmp_path / "a.ppm"
    assert lib.rtmi_write_ppm(str(p).encode(), img.ctypes.data, w, h, 0) == 0
    raw = p.read_bytes()
    assert raw.startswith(b"P6\n3 2\n255\n")
    px = np.frombuffer(raw[len(b"P6\n3 2\n255\n"):], dtype=np.uint8).reshape(h, w, 3)
    assert np.array_equal(px[0, :, 1], [100, 150, 200]) and np.array_equal(px[1, :, 0], [10, 20, 30])
    assert lib.rtmi_write_ppm(str(p).encode(), img.ctypes.data, w, h, 1) == 0
    px = np.frombuffer(p.read_bytes()[len(b"P6\n3 2\n255\n"):], dtype=np.uint8).reshape(h, w, 3)

    def srgb(v):
        c = v / 255.0
        s = 12.92 * c if c <= 0.0031308 else 1.055 * c ** (1 / 2.4) - 0.055
        return int(round(s * 255))
    assert list(px[0, :, 1]) == [srgb(100), srgb(150), srgb(200)] and px[0, 0, 0] == 0
    acc = np.arange(h * w * 4, dtype=np.float32).reshape(h, w, 4)
    q = tmp_path / "a.pfm"
    assert lib.rtmi_write_pfm(str(q).encode(), acc.ctypes.data, w, h, 4) == 0
    raw = q.read_bytes()
    head = b"PF\n3 2\n-1.0\n"
    assert raw.startswith(head)
    data = np.frombuffer(raw[len(head):], dtype="<f4").reshape(h, w, 3)
    assert np.array_equal(data, acc[..., :3] / 4.0)
    assert lib.rtmi_write_ppm(b"/nonexistent_dir/x.ppm", img.ctypes.data, w, h, 0) == rtmi.abi.ERR_IO


def _read_exr(raw):
    """A reader for the plain form of OpenEXR (one part, scan lines, no compression), written from the
    format's file layout: returns (attributes, {channel name: [h, w] float32 array, top row first})."""
    import struct
    assert struct.unpack_from("<i", raw, 0)[0] == 20000630
    version = struct.unpack_from("<I", raw, 4)[0]
    assert version & 0xFF == 2 and version >> 8 == 0          # version 2, no tiles / long names / deep data / parts
    pos, attrs = 8, {}
    while raw[pos] != 0:
        end = raw.index(b"\0", pos)
        name = raw[pos:end].decode()
        pos = end + 1
        end = raw.index(b"\0", pos)
        kind = raw[pos:end].decode()
        pos = end + 1
        size = struct.unpack_from("<i", raw, pos)[0]
        attrs[name] = (kind, raw[pos + 4:pos + 4 + size])
        pos += 4 + size
    pos += 1
    kind, box = attrs["dataWindow"]
    assert kind == "box2i"
    x0, y0, x1, y1 = struct.unpack("<4i", box)
    w, h = x1 - x0 + 1, y1 - y0 + 1
    channels, c = [], 0
    blob = attrs["channels"][1]
    while blob[c] != 0:
        end = blob.index(b"\0", c)
        pixel_type, linear, xs, ys = struct.unpack_from("<iB3xii", blob, end + 1)
        channels.append((blob[c:end].decode(), pixel_type, xs, ys))
        c = end + 1 + 16
    assert c == len(blob) - 1
    offsets = struct.unpack_from("<%dQ" % h, raw, pos)
    assert offsets[0] == pos + 8 * h
    out = {name: np.zeros((h, w), dtype=np.float32) for name, _, _, _ in channels}
    for row, off in enumerate(offsets):
        y, nbytes = struct.unpack_from("<ii", raw, off)
        assert y == y0 + row and nbytes == 4 * w * len(channels)
        for k, (name, pixel_type, xs, ys) in enumerate(channels):
            assert pixel_type == 2 and xs == 1 and ys == 1
            out[name][row] = np.frombuffer(raw, dtype="<f4", count=w, offset=off + 8 + 4 * w * k)
    assert offsets[-1] + 8 + 4 * w * len(channels) == len(raw)
    return attrs, out


def test_exr_writer(rtmi, tmp_path):
    lib = rtmi.abi.load()
    w, h, frames = 5, 3, 7
    rs = np.random.RandomState(3)
    acc = rs.uniform(0.0, 40.0, size=(h, w, 4)).astype(np.float32)
    acc[1, 2, :3] = [np.inf, 0.0, 1e-40]
    p = tmp_path / "a.exr"
    assert lib.rtmi_write_exr(str(p).encode(), acc.ctypes.data, w, h, frames) == 0
    attrs, ch = _read_exr(p.read_bytes())
    assert sorted(ch) == ["B", "G", "R"]
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    assert attrs["displayWindow"] == attrs["dataWindow"]
    assert attrs["pixelAspectRatio"][1] == np.float32(1.0).tobytes() and attrs["screenWindowWidth"][1] == np.float32(1.0).tobytes()
    assert attrs["screenWindowCenter"] == ("v2f", bytes(8))
    want = acc[::-1, :, :3] / np.float32(frames)                 # top row first; buffer row 0 is the bottom of the view
    for k, name in enumerate("RGB"):
        assert np.array_equal(ch[name].view(np.uint32), want[..., k].view(np.uint32))
    assert lib.rtmi_write_exr(b"/nonexistent_dir/x.exr", acc.ctypes.data, w, h, frames) == rtmi.abi.ERR_IO
    assert lib.rtmi_write_exr(str(p).encode(), None, w, h, frames) == rtmi.abi.ERR_INVALID_ARGUMENT


def _records(sc, rtmi):
    return (bytes(sc.sphere_array())[:48 * sc.n_spheres], bytes(sc.triangles)[:96 * sc.n_triangles],
            bytes(sc.meshes)[:80 * sc.n_meshes])


def test_scene_file_round_trip_is_bit_exact(rtmi, tmp_path):
    """Every float of every record comes back with the same bits, including values that need all nine
    digits, denormals, negative zero and infinities."""
    sc, _ = rtmi.scenes.stress_scene(200, (6, 5), 3, seed=99, extent=3.0)
    sph = np.frombuffer(sc.sphere_array(), dtype=rtmi.host.SPHERE_DTYPE, count=sc.n_spheres).copy()
    sph["center"][0] = [np.float32(1.0) / np.float32(3.0), -0.0, np.float32(1e-42)]
    sph["radius"][1] = np.inf
    sph["center"][2] = [np.float32(16777217.0), np.nextafter(np.float32(1), np.float32(2)), np.float32(-3.4028235e38)]
    tri = np.frombuffer(sc.triangles, dtype=rtmi.host.TRIANGLE_DTYPE, count=sc.n_triangles)
    msh = np.frombuffer(sc.meshes, dtype=rtmi.host.MESH_DTYPE, count=sc.n_meshes)
    sc = rtmi.Scene.from_records(sph, tri, msh)
    path = tmp_path / "scene.txt"
    sc.save(path)
    back = rtmi.Scene.load(path)
    assert (back.n_spheres, back.n_triangles, back.n_meshes) == (200, 60, 3)
    assert _records(back, rtmi) == _records(sc, rtmi)
    # materials of the transmitting-material extension keep their two extra words
    glass, _ = rtmi.scenes.stress_scene(40, (4, 4), 4, seed=5, extent=3.0, glass=True)
    glass.save(path)
    assert _records(rtmi.Scene.load(path), rtmi) == _records(glass, rtmi)
    assert "1.5" in path.read_text()
    # the reference scene too, and an empty scene
    ref = rtmi.reference_scene()
    ref.save(path)
    assert _records(rtmi.Scene.load(path), rtmi) == _records(ref, rtmi)
    rtmi.Scene().save(path)
    empty = rtmi.Scene.load(path)
    assert (empty.n_spheres, empty.n_triangles, empty.n_meshes) == (0, 0, 0)


def test_scene_file_is_readable_text_and_rejects_damage(rtmi, tmp_path):
    path = tmp_path / "s.txt"
    path.write_text("# hand-written\nrtmi-scene 1\n\nsphere 0 1 2 0.5  1 0.5 0.25 2 0.75\n"
                    "tri 0 0 0  1 0 0  0 1 0   0 0 1  0 0 1  0 0 1\n"
                    "mesh 0 1  0 0 0  1 1 0  0.9 0.9 0.9 0 0.1\n")
    sc = rtmi.Scene.load(path)
    s = sc.sphere_array()[0]
    assert list(s.center) == [0.0, 1.0, 2.0] and s.radius == 0.5 and s.material.light == 2.0 and s.material.smoothness == 0.75
    assert list(sc.triangles[0].p2)[:3] == [1.0, 0.0, 0.0] and list(sc.triangles[0].n3)[:3] == [0.0, 0.0, 1.0]
    m = sc.meshes[0]
    assert (m.start_triangle, m.num_triangles) == (0, 1) and list(m.bounding_point2)[:3] == [1.0, 1.0, 0.0]
    good = path.read_text()
    for bad in (good.replace("rtmi-scene 1", "rtmi-scene 2"),       # unknown version
                good.replace("rtmi-scene 1\n", ""),                 # no header
                good.replace("0.75\n", "0.75 9\n"),                 # extra field
                good.replace("sphere 0 1 2 0.5", "sphere 0 1 2"),   # missing field
                good.replace("mesh 0 1", "mesh 0 2"),               # range beyond the triangles
                good + "cone 1 2 3\n"):                             # unknown record
        path.write_text(bad)
        with pytest.raises(rtmi.RtmiError) as e:
            rtmi.Scene.load(path)
        assert e.value.code == rtmi.abi.ERR_IO
    with pytest.raises(rtmi.RtmiError):
        rtmi.Scene.load(tmp_path / "missing.txt")

