"""Render a workload's image and write it as PNG (display chain of the reference: sum / frames, 8-bit,
sRGB encode, rows flipped so that the picture is upright).  usage: render_png.py workload out.png [frames] [w h]"""
import os, struct, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as R

name, out = sys.argv[1], sys.argv[2]
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 64
w, h = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (640, 360)
if name == "stress":
    scene, pose = R.scenes.stress_scene()
    depth = 16
elif name == "stress_small":
    scene, pose = R.scenes.stress_scene(5000, (50, 50), 4, extent=5.0)
    depth = 16
else:
    scene, pose, depth = (R.sphere_scene() if name == "spheres" else R.reference_scene()), R.BENCH_POSE, 8
fd = R.make_frame_data(w, h, scene, pose, 1, depth)
with R.Renderer(w, h) as r:
    r.upload_scene(scene)
    r.render_frames(fd, 1, frames)
    rgba = r.read_output_rgba8()
lin = rgba[..., :3].astype(np.float64) / 255.0
srgb = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * lin ** (1 / 2.4) - 0.055)
img = (np.clip(srgb, 0, 1) * 255.0 + 0.5).astype(np.uint8)[::-1]           # buffer row 0 is the bottom of the view
raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))
def chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
open(out, "wb").write(png)
print("wrote", out, img.shape, "mean", img.mean(axis=(0, 1)))
