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
    """Preallocated gather of per-rank packed tiles to `dst` + un-interleave into the frame."""

    def __init__(self, n_rays: int, tile_rays: int, channels: int, device, dtype=torch.float32, group=None, dst=0):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.dst = dst
        self.n_rays, self.tile_rays, self.channels = n_rays, tile_rays, channels
        self.max_local = sharding.max_local_rays(n_rays, tile_rays, self.world)
        self.local_rays = sharding.local_rays(n_rays, tile_rays, self.rank, self.world)
        shape = (self.max_local, channels) if channels > 1 else (self.max_local,)
        # every rank sends max_local rows (ranks with one tile fewer leave the tail unused)
        self.local = torch.zeros(shape, dtype=dtype, device=device)
        self.recv = [torch.empty_like(self.local) for _ in range(self.world)] if self.rank == dst and self.world > 1 else None

    def gather(self):
        """Returns the assembled frame on rank `dst`, None elsewhere."""
        if self.world == 1:
            return self.local[: self.n_rays]
        backend = dist.get_backend(self.group)
        if backend == "gloo" and self.local.is_cuda:
            # rehearsal mode (several ranks sharing one GPU, no RCCL): stage through host memory
            send = self.local.cpu()
            if self.rank == self.dst:
                recv = [torch.empty_like(send) for _ in range(self.world)]
                dist.gather(send, recv, dst=self.dst, group=self.group)
                return sharding.assemble_frame([r.to(self.local.device) for r in recv], self.tile_rays, self.n_rays)
            dist.gather(send, None, dst=self.dst, group=self.group)
            return None
        if self.rank == self.dst:
            dist.gather(self.local, self.recv, dst=self.dst, group=self.group)
            return sharding.assemble_frame(self.recv, self.tile_rays, self.n_rays)
        dist.gather(self.local, None, dst=self.dst, group=self.group)
        return None


class ShardedHIPRaytracer:
    """IRaytracer-shaped front for N ranks: Render() returns the full frame on rank 0 (None elsewhere)."""

    def __init__(self, objects, lights, rays, MAX_BOUNCES, *, camera=None, kernel="shade_and_reflect",
                 tile_rows: int = 16, width: int | None = None, device_index: int = 0, group=None, **kw):
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
        self.gatherer = FrameGather(self.n_rays, self.tile_rays, self.rt.elem_floats, self.device, group=group)
        assert self.gatherer.local_rays == self.rt.local_rays

    def render_local(self):
        """Asynchronous: this rank's tiles into its torch buffer, on torch's current stream."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.rt.render_device(self.gatherer.local.data_ptr(), stream)

    def Render(self):
        self.render_local()
        return self.gatherer.gather()

    def close(self):
        self.rt.close()
