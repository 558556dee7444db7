"""RCCL on the image the renderer is bound to, with one rank (a one-GPU box cannot hold more): the
kernels are launched on a torch stream through the C ABI, the all-reduce follows without a host
synchronisation, and the result must be the plain render (a sum over one rank).  Prints OK or FAIL."""
import os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MASTER_ADDR"] = "127.0.0.1"
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
import numpy as np, torch, torch.distributed as dist
import raytracer_amd as R
from raytracer_amd.distributed import ShardedRenderer
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
w, h = 320, 180
scene = R.sphere_scene()
fd = R.make_frame_data(w, h, scene, R.BENCH_POSE, 1, 8)
r = R.Renderer(w, h, device=0, variant=2); r.upload_scene(scene)
sr = ShardedRenderer(r, dev)
for step in range(3):
    sr.accum.fill_(5.0)                                     # junk, on torch's default stream
    sr.stream.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(sr.stream):                      # the stream the library launches on
        sr.accum.zero_()
        r.render_frames(fd, 1, 16)
        dist.all_reduce(sr.accum, op=dist.ReduceOp.SUM)     # RCCL on the tensor the kernels wrote, no host sync in between
    sr.resolve(16)
torch.cuda.synchronize()
got = sr.accum.cpu().numpy()
with R.Renderer(w, h, device=0, variant=2) as r2:
    r2.upload_scene(scene); r2.render_frames(fd, 1, 16); want = r2.read_accum()
ok = bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
print("OK" if ok else "FAIL", "RCCL all_reduce on the bound image vs a plain render")
dist.destroy_process_group()
sys.exit(0 if ok else 1)
