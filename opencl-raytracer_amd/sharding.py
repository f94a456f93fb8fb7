"""Row-tile partition of the ray grid across GPUs, and assembly of the gathered framebuffer.

The reference is single-device (OpenCLRaytracer.cpp:36-44). Pixels are independent, so the frame is cut into
tiles of `tile_rays` consecutive rays (tile_rows * width), tile j owned by rank j % world (interleaved, because
cost per row varies with hit fraction - SURVEY.md 8e). Each rank renders its tiles packed back to back
(rt_set_shard); the only exchange is one gather of framebuffer tiles to rank 0 (RCCL over xGMI with the
"nccl" backend; "gloo" in CPU tests). No collective sits inside the data path of a render.
"""
from __future__ import annotations

import numpy as np


def n_tiles(n_rays: int, tile_rays: int) -> int:
    return (n_rays + tile_rays - 1) // tile_rays


def local_tiles(n_rays: int, tile_rays: int, rank: int, world: int) -> list[int]:
    return list(range(rank, n_tiles(n_rays, tile_rays), world))


def local_rays(n_rays: int, tile_rays: int, rank: int, world: int) -> int:
    """Work-items rank renders (whole tiles; the ragged last tile is padded) - equals rt_local_rays()."""
    if world <= 1:
        return n_rays
    return len(local_tiles(n_rays, tile_rays, rank, world)) * tile_rays


def max_local_rays(n_rays: int, tile_rays: int, world: int) -> int:
    return max(local_rays(n_rays, tile_rays, r, world) for r in range(world))


def tile_rays_for_rows(width: int, tile_rows: int) -> int:
    return width * tile_rows


def assemble_frame(pieces, tile_rays: int, n_rays: int):
    """Un-interleave the per-rank packed buffers (index = rank) into the full frame. Works on numpy arrays
    and torch tensors alike (CPU or GPU): pieces[r] has shape (>= local_rays(r), C...)."""
    world = len(pieces)
    first = pieces[0]
    if world == 1:
        return first[:n_rays]
    is_torch = not isinstance(first, np.ndarray)
    tiles = n_tiles(n_rays, tile_rays)
    shape = (tiles * tile_rays,) + tuple(first.shape[1:])
    if is_torch:
        import torch
        frame = torch.empty(shape, dtype=first.dtype, device=first.device)
    else:
        frame = np.empty(shape, dtype=first.dtype)
    view = frame.reshape((tiles, tile_rays) + tuple(first.shape[1:]))
    for r, piece in enumerate(pieces):
        mine = len(range(r, tiles, world))
        if mine == 0:
            continue
        src = piece[: mine * tile_rays].reshape((mine, tile_rays) + tuple(first.shape[1:]))
        view[r::world] = src
    return frame[:n_rays]
