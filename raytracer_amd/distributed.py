"""Multi-GPU rendering: one process per GPU, frames split by contiguous range, one sum-reduce of the
float4 accumulation images at the end.

A frame's contribution to a pixel depends only on (x, y, frame_index, scene, camera): the seed is a
pure function of them (Raytracing.comp:462) and nothing is read back from earlier frames except the
running sum (Raytracing.comp:471-472).  The samples inside one frame share an RNG stream
(Raytracing.comp:464-468), so the smallest unit that can move to another GPU is a whole frame.
Every rank therefore renders its own frame range into a zeroed local image (adding into zero equals
the shader's frame_index == 1 overwrite) and the images are summed with one collective
(torch.distributed: RCCL over xGMI on GPUs, gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def split_frames(total_frames, world_size, rank, first_frame=1):
    """Contiguous, balanced frame ranges: returns (first_frame_of_rank, n_frames_of_rank)."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank %d / world size %d" % (rank, world_size))
    q, rem = divmod(int(total_frames), world_size)
    n = q + (1 if rank < rem else 0)
    start = first_frame + rank * q + min(rank, rem)
    return start, n


def render_sharded(render_range, accum, total_frames, first_frame=1, group=None, dst=None, around_collective=None):
    """Render this rank's share of frames first_frame .. first_frame+total_frames-1 into `accum`
    (a [H, W, 4] float32 tensor the renderer is bound to) and sum the images over all ranks.

    render_range(first, n) must add frames first .. first+n-1 into `accum`, overwriting instead when
    first == 1, exactly as rtmi_render_frames does.  With dst=None every rank ends up with the sum
    (all-reduce); otherwise only rank dst does (reduce).  Returns (first, n) of this rank.  The
    alpha channel of the sum equals the number of ranks that rendered something; the caller puts
    it back to 1 (rtmi_resolve_output does).  around_collective, if given, is called with "before" and
    "after" on either side of the collective (the bench records events there)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    first, n = split_frames(total_frames, world, rank, first_frame)
    if world > 1 or first != 1 or n == 0:
        # ranks other than the owner of frame 1 add into their buffer, so it has to start at zero;
        # the owner of frame 1 overwrites, but is zeroed too so every rank runs the same sequence
        accum.zero_()
    if n > 0:
        render_range(first, n)
    if world > 1:
        if accum.is_cuda and dist.get_backend(group) == "gloo":
            # gloo (the CPU tests' backend, and the one-GPU rehearsal of the multi-GPU bench) stages device
            # tensors through the host on streams of its own; started while this rank's render kernel is still
            # running, with two processes sharing one GPU, it has been seen to stall the kernel for seconds or
            # for good.  RCCL, the backend that matters, orders itself after the kernels on the device.
            torch.cuda.current_stream(accum.device).synchronize()
        if around_collective is not None:
            around_collective("before")
        if dst is None:
            dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
        if around_collective is not None:
            around_collective("after")
    return first, n


class ShardedRenderer:
    """A Renderer bound to a torch tensor so that torch.distributed can reduce the accumulation
    image in place.

    Everything that touches the image (clearing it, the render kernels, the collective, the resolve
    kernel) is enqueued on ONE torch stream owned by this object, whose non-zero handle the library
    launches on (rtmi_set_stream treats a NULL handle as "the context's own stream", so torch's
    default stream, handle 0, cannot be named that way).  The stream is joined to the caller's
    current stream on entry and exit, so the whole sequence is ordered by the device alone, without
    a host synchronisation."""

    def __init__(self, renderer, device):
        self.renderer = renderer
        self.device = torch.device(device)
        self.accum = torch.zeros((renderer.height, renderer.width, 4), dtype=torch.float32, device=self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        assert self.stream.cuda_stream != 0
        # the zero fill above ran on the caller's stream
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        renderer.bind_accum(self.accum.data_ptr())
        renderer.set_stream(self.stream.cuda_stream)
        self._collective_events = None   # time_collectives(True): one (before, after) event pair per render()

    def time_collectives(self, enable=True):
        """Record an event pair on this object's stream around every collective from now on (the events
        cost no synchronisation); collective_ms() reads them."""
        self._collective_events = [] if enable else None

    def collective_ms(self):
        """(summed milliseconds, count) between the event pairs recorded since time_collectives(True): on this
        rank, the time from the end of its own render kernels to the completion of the collective, i.e. the
        transfer plus the wait for slower ranks.  Waits for the events."""
        pairs = self._collective_events or []
        total = 0.0
        for before, after in pairs:
            after.synchronize()
            total += before.elapsed_time(after)
        n = len(pairs)
        if self._collective_events is not None:
            self._collective_events = []
        return total, n

    def render(self, fd, total_frames, first_frame=1, group=None, dst=None, join=True):
        """This rank's share of the frames + the sum over ranks (all ranks, or rank dst only).

        join=False leaves the caller's stream out of it at the end: the sequence stays on this object's stream
        and the caller synchronises when it needs the image (join() below, or a device-wide wait).  That is what
        lets two ShardedRenderers, each with a context and an image of its own, take alternate images of a
        sequence: the collective of one image then runs while the other context renders the next."""
        caller = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(caller)
        hook = None
        if self._collective_events is not None:
            pair = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]

            def hook(which):
                pair[0 if which == "before" else 1].record(self.stream)
                if which == "after":
                    self._collective_events.append(tuple(pair))
        with torch.cuda.stream(self.stream):
            rng = render_sharded(lambda first, n: self.renderer.render_frames(fd, first, n), self.accum,
                                 total_frames, first_frame, group, dst, hook)
        if join:
            caller.wait_stream(self.stream)
        return rng

    def resolve(self, total_frames, join=True):
        """Alpha back to 1 and the display image = sum / total_frames (on the rank(s) holding the sum)."""
        caller = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(caller)
        self.renderer.resolve_output(total_frames)   # launches on self.stream
        if join:
            caller.wait_stream(self.stream)

    def join(self):
        """The caller's current stream waits for everything enqueued here so far (after render / resolve with join=False)."""
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def close(self):
        """Give the renderer back its own image and stream."""
        self.renderer.synchronize()
        self.renderer.bind_accum(None)
        self.renderer.set_stream(None)
