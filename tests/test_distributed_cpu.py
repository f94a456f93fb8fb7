"""CPU, world_size 2 over gloo: the N>1 path (interleaved row-tile shards + one gather to rank 0 + assembly).

The per-rank render is stood in for by the CPU oracle (tests may use it); the partition arithmetic, the
collective and the un-interleave are exactly the code the GPU path runs (distributed.FrameGather)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, tile_rows, W, H, result_path):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import camera, random_scene
        from opencl_raytracer_amd import sharding
        from opencl_raytracer_amd.distributed import FrameGather
        from oracle import oracle
        objs, lights = random_scene(6, 4, 2, seed=77)
        n_rays = W * H
        tile = sharding.tile_rays_for_rows(W, tile_rows)
        fg = FrameGather(n_rays, tile, 4, torch.device("cpu"))
        assert fg.local_rays == sharding.local_rays(n_rays, tile, rank, world)
        rs = oracle.Restatement(True)
        rays = camera.primary_rays(W, H)
        # render my tiles, packed back to back (what rt_set_shard does on the GPU)
        off = 0
        for t in sharding.local_tiles(n_rays, tile, rank, world):
            chunk = rays[t * tile:(t + 1) * tile]
            out = rs.render("shade_and_reflect", objs, lights, chunk, 2, threads=1)["out"]
            fg.local[off:off + len(chunk)] = torch.from_numpy(out)
            off += tile
        frame = fg.gather()
        if rank == 0:
            want = rs.render("shade_and_reflect", objs, lights, rays, 2, threads=1)["out"]
            ok = np.array_equal(frame.numpy().view(np.uint32), want.view(np.uint32))
            Path(result_path).write_text("ok" if ok else "mismatch")
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tile_rows,W,H", [(2, 2, 16, 12), (2, 4, 16, 10), (3, 2, 16, 14), (3, 8, 16, 12)])
def test_gather_assembles_the_frame(tmp_path, world, tile_rows, W, H):
    """even split; ragged last tile + uneven ranks; three ranks; a rank with no tile at all (2 tiles, 3 ranks)"""
    port = 29500 + (os.getpid() % 2000) + tile_rows + 7 * world
    result = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, port, tile_rows, W, H, str(result)), nprocs=world, join=True)
    assert result.read_text() == "ok"
