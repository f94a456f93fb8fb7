"""Worker of tests/test_distributed_gpu.py: one rank of an N-rank render + gather, launched by torch.distributed.run.

    argv[1] = backend: "nccl" (one GPU per rank, RCCL over xGMI) or "gloo" (rehearsal: every rank on cuda:0, messages
    staged through host memory - what a one-GPU box can run).
Rank 0 checks the gathered frame bit for bit against a single-context render of the whole frame and exits non-zero on a
mismatch."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    backend = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group("gloo")
    from helpers import camera
    from opencl_raytracer_amd import synthetic
    from opencl_raytracer_amd.distributed import ShardedHIPRaytracer
    from opencl_raytracer_amd.hip_raytracer import HIPRaytracer
    ok = True
    # small-scene kernel / grid path; even split / ragged last tile with uneven tile counts
    for n_objs, (W, H, tile_rows) in ((12, (256, 128, 16)), (600, (256, 128, 16)), (600, (192, 104, 8))):
        objs, lights = synthetic.spheres_and_lights(n_objs, 3)
        z = float(camera.camera_z(H))
        want = None
        if rank == 0:
            with HIPRaytracer(objs, lights, None, 3, camera=(W, H, z), device=local_rank) as rt:
                want = rt.Render()
        for pipeline in (False, True):
            # pipelined: frame k's exchange overlaps frame k + 1's render (two slots of every buffer, three streams); five
            # frames, every one checked, every buffer poisoned with NaN before it is reused
            srt = ShardedHIPRaytracer(objs, lights, None, 3, camera=(W, H, z), tile_rows=tile_rows, device_index=local_rank,
                                      pipeline=pipeline)
            srt.gatherer.debug_poison = pipeline
            same = True
            for _ in range(5 if pipeline else 2):
                frame = srt.Render()
                if rank == 0:
                    got = frame.cpu().numpy()   # (on the current stream: ordered behind the frame's assembly)
                    same = same and got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
                else:
                    assert frame is None
            if rank == 0:
                print(f"[rank 0] {backend} world {world} N={n_objs} {W}x{H} tile_rows {tile_rows} pipeline {pipeline}: "
                      f"{'ok' if same else 'MISMATCH'}", flush=True)
                ok = ok and same
            srt.close()
            dist.barrier()
    flag = torch.tensor([1 if ok else 0])
    if backend == "nccl":
        flag = flag.cuda()
    dist.broadcast(flag, src=0)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
