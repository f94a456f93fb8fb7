"""Multi-GPU rendering: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm).

The scene is replicated (each rank uploads the same record buffers), the ray grid is partitioned into
interleaved row-tiles (sharding.py), every rank renders its tiles with its own HIPRaytracer into a torch
tensor, and ONE gather per frame assembles the framebuffer on rank 0. There is no exchange between bounces
and no all-reduce anywhere. torch is plumbing here (device memory, streams, the collective);
the render itself goes through the C ABI with raw pointers.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import sharding


class FrameGather:
    """Per-frame exchange: every rank's packed tiles to `dst`, where they land in the frame at their final offsets.

    One exact-size message per peer (grouped isend / irecv = one RCCL group; each peer has its own xGMI link to the
    root, so the seven transfers of an 8-GPU node run in parallel: 33.5 MB each at 4096^2) into a per-peer staging
    buffer, then one strided device copy per peer into the interleaved tile slots of the frame - no padding to the
    largest share, no intermediate list of max-size tensors. `dst` copies its own tiles while the messages are in
    flight. Nothing is posted before the local render has been enqueued: the receive kernels of an early post would
    occupy compute units of the root for the whole render, and with interleaved tiles the ranks finish together anyway.
    """

    def __init__(self, n_rays: int, tile_rays: int, channels: int, device, dtype=torch.float32, group=None, dst=0,
                 pipeline: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.dst = dst
        self.n_rays, self.tile_rays, self.channels = n_rays, tile_rays, channels
        self.tiles = sharding.n_tiles(n_rays, tile_rays)
        self.local_rays = sharding.local_rays(n_rays, tile_rays, self.rank, self.world)
        tail = (channels,) if channels > 1 else ()
        device = torch.device(device)
        # Pipelined mode (GPU, world > 1): every buffer exists twice and frame k's exchange runs while frame k + 1 is being
        # rendered - see exchange_pipelined(). Slot = frame number mod 2.
        self.pipeline = bool(pipeline) and self.world > 1 and device.type == "cuda"
        n_slots = 2 if self.pipeline else 1
        self.locals = [torch.zeros((self.local_rays,) + tail, dtype=dtype, device=device) for _ in range(n_slots)]
        self.frames = [None] * n_slots
        self.stagings = [dict() for _ in range(n_slots)]
        if self.rank == dst and self.world > 1:
            for slot in range(n_slots):
                # the frame in whole tiles (the ragged last tile is padded; the caller gets frame[:n_rays])
                self.frames[slot] = torch.empty((self.tiles * tile_rays,) + tail, dtype=dtype, device=device)
                for r in range(self.world):
                    if r != dst:
                        n = sharding.local_rays(n_rays, tile_rays, r, self.world)
                        self.stagings[slot][r] = torch.empty((n,) + tail, dtype=dtype, device=device)
        self.k = 0                       # frames started (pipelined mode)
        self.pending = [None] * n_slots  # the exchange that last read locals[slot] / wrote stagings[slot]
        self.placed = [None] * n_slots   # dst: event after which frames[slot] is complete and locals / stagings[slot] are free again
        self.keep = [None] * n_slots     # (rehearsal over gloo: the host copy a send is reading)
        self.debug_poison = False        # tests: overwrite every buffer with NaN before it is reused, so that stale data cannot pass for a frame
        self.render_stream = torch.cuda.Stream(device) if self.pipeline else None
        self.place_stream = torch.cuda.Stream(device) if (self.pipeline and self.rank == dst) else None

    # (the one-slot names the synchronous path and its callers use)
    @property
    def local(self):
        return self.locals[self.k % len(self.locals)]

    @property
    def frame(self):
        return self.frames[0]

    @property
    def staging(self):
        return self.stagings[0]

    def _place(self, piece, r, frame=None):
        """rank r's packed tiles -> tile slots r, r + world, ... of the frame (one strided copy)"""
        frame = self.frame if frame is None else frame
        mine = len(range(r, self.tiles, self.world))
        if mine == 0:
            return
        tail = tuple(frame.shape[1:])
        view = frame.view((self.tiles, self.tile_rays) + tail)
        view[r::self.world] = piece[: mine * self.tile_rays].view((mine, self.tile_rays) + tail)

    # ---- pipelined mode ---------------------------------------------------------------------------------------------
    # Frame k lives in slot k % 2. Three streams: the context's RENDER stream (all renders, in order), on `dst` a PLACE
    # stream (assembly of one frame), and the caller's current stream (the consumer). Per frame:
    #   begin_frame()  render stream waits until slot's buffers are free: the send that read locals[slot] two frames ago
    #                  (peers), the assembly that read locals / stagings[slot] two frames ago (dst);
    #   [the caller renders into self.local on self.render_stream]
    #   exchange_pipelined()  peers: post the send behind the render, do NOT wait for it; dst: post the receives, then on the
    #                  place stream - own tiles, wait for the messages, the peers' tiles - and make the CALLER's stream (not
    #                  the render stream) wait for that. The next frame's render is enqueued behind the previous RENDER only,
    #                  so it runs while this frame's tiles travel and are put in place.
    # The frame returned for call k is overwritten by call k + 2.
    def begin_frame(self):
        slot = self.k % 2
        with torch.cuda.stream(self.render_stream):
            for req in self.pending[slot] or ():
                req.wait()
            self.pending[slot] = None
            if self.placed[slot] is not None:
                self.render_stream.wait_event(self.placed[slot])
            if self.debug_poison:
                self.locals[slot].fill_(float("nan"))
                for buf in self.stagings[slot].values():
                    buf.fill_(float("nan"))
        return self.render_stream

    def exchange_pipelined(self):
        slot = self.k % 2
        self.k += 1
        local = self.locals[slot]
        staged_through_host = dist.get_backend(self.group) == "gloo"
        if self.rank != self.dst:
            if self.local_rays:
                with torch.cuda.stream(self.render_stream):   # (the send is ordered behind the render; .cpu() waits for it)
                    send = local.cpu() if staged_through_host else local
                    self.keep[slot] = send
                    self.pending[slot] = dist.batch_isend_irecv([dist.P2POp(dist.isend, send, self.dst, self.group)])
            return None
        frame, staging = self.frames[slot], self.stagings[slot]
        recv = {r: (torch.empty(buf.shape, dtype=buf.dtype) if staged_through_host else buf)
                for r, buf in staging.items() if buf.shape[0]}
        with torch.cuda.stream(self.render_stream):  # (behind begin_frame's waits: stagings[slot] is free)
            reqs = dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, r, self.group) for r, buf in recv.items()]) if recv else []
        consumer = torch.cuda.current_stream(local.device)
        self.place_stream.wait_stream(consumer)             # whoever still reads frames[slot] (two frames old)
        self.place_stream.wait_stream(self.render_stream)   # own tiles rendered
        with torch.cuda.stream(self.place_stream):
            if self.debug_poison:
                frame.fill_(float("nan"))
            self._place(local, self.dst, frame)
            for req in reqs:
                req.wait()
            for r, buf in recv.items():
                self._place(buf.to(frame.device, non_blocking=False) if staged_through_host else buf, r, frame)
            done = torch.cuda.Event()
            done.record(self.place_stream)
        self.placed[slot] = done
        consumer.wait_event(done)
        return frame[: self.n_rays]

    def drain(self):
        """Pipelined mode: wait (host) until nothing of this rank's exchanges is in flight."""
        if not self.pipeline:
            return
        for slot in range(2):
            for req in self.pending[slot] or ():
                req.wait()
            self.pending[slot] = None
        # (over RCCL req.wait() only makes the CURRENT stream wait for the transfer: a host-side guarantee needs the
        #  device-wide synchronisation - a peer's last isend may still be in flight behind the two streams above)
        torch.cuda.synchronize(self.render_stream.device)

    def gather(self):
        """Returns the assembled frame on rank `dst`, None elsewhere."""
        if self.world == 1:
            return self.local[: self.n_rays]
        staged_through_host = dist.get_backend(self.group) == "gloo" and self.local.is_cuda
        if self.rank != self.dst:
            if self.local_rays:
                # rehearsal mode (several ranks sharing one GPU over gloo, no RCCL): the message goes through host memory
                send = self.local.cpu() if staged_through_host else self.local
                for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, send, self.dst, self.group)]):
                    req.wait()
            return None
        recv = {r: (torch.empty(buf.shape, dtype=buf.dtype) if staged_through_host else buf)
                for r, buf in self.staging.items() if buf.shape[0]}
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, r, self.group) for r, buf in recv.items()]) if recv else []
        self._place(self.local, self.dst)   # own tiles, while the peers' are in flight
        for req in reqs:
            req.wait()
        for r, buf in recv.items():
            self._place(buf.to(self.frame.device) if staged_through_host else buf, r)
        return self.frame[: self.n_rays]


class ShardedHIPRaytracer:
    """IRaytracer-shaped front for N ranks: Render() returns the full frame on rank 0 (None elsewhere)."""

    def __init__(self, objects, lights, rays, MAX_BOUNCES, *, camera=None, kernel="shade_and_reflect",
                 tile_rows: int = 16, width: int | None = None, device_index: int = 0, group=None,
                 pipeline: bool = False, **kw):
        from .hip_raytracer import HIPRaytracer
        self.rt = HIPRaytracer(objects, lights, rays, MAX_BOUNCES, kernel=kernel, device=device_index,
                               camera=camera, **kw)
        self.n_rays = self.rt.n_rays
        w = width if width is not None else (camera[0] if camera is not None else self.rt.stats().width)
        if not w:
            raise ValueError("width is required to cut row tiles when the rays are not a pinhole grid")
        self.tile_rays = sharding.tile_rays_for_rows(int(w), tile_rows)
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rt.set_shard(self.tile_rays, rank, world)
        self.device = torch.device("cuda", device_index)
        self.gatherer = FrameGather(self.n_rays, self.tile_rays, self.rt.elem_floats, self.device, group=group, pipeline=pipeline)
        assert self.gatherer.local_rays == self.rt.local_rays  # exact share: nothing is padded to the largest one

    def render_local(self):
        """Asynchronous: this rank's tiles into its torch buffer, on torch's current stream."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.rt.render_device(self.gatherer.local.data_ptr(), stream)

    def Render(self):
        g = self.gatherer
        if not g.pipeline:
            self.render_local()
            return g.gather()
        # pipelined: this frame's render starts as soon as the previous RENDER is done - while the previous frame's tiles are
        # still travelling to rank 0 and being put in place (FrameGather.exchange_pipelined). The returned frame is valid in
        # the caller's stream order and is overwritten by the second-next Render().
        stream = g.begin_frame()
        self.rt.render_device(g.local.data_ptr(), stream.cuda_stream)
        return g.exchange_pipelined()

    def render_synchronous(self):
        """One frame through the synchronous path (render, then exchange, then assembly, all on the caller's stream),
        whatever mode the object is in; returns a COPY of the frame on rank 0. For self-checks of the pipelined mode."""
        g = self.gatherer
        g.drain()
        torch.cuda.synchronize(self.device)
        saved, k0 = g.pipeline, g.k
        g.pipeline, g.k = False, 0
        try:
            self.render_local()
            frame = g.gather()
            torch.cuda.synchronize(self.device)
            return None if frame is None else frame.clone()
        finally:
            g.pipeline, g.k = saved, k0 + (k0 & 1)   # (continue with slot 0: nothing is in flight)

    def close(self):
        self.gatherer.drain()
        self.rt.close()
