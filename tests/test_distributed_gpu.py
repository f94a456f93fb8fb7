"""GPU, N ranks: interleaved row-tile shards rendered through the C ABI, one exchange to rank 0, frame equal to a
single-context render bit for bit.

* backend "gloo": two and three ranks SHARING the one GPU of the test box (messages staged through host memory) - runs
  everywhere a GPU is;
* backend "nccl": one GPU per rank over RCCL / xGMI - skipped unless the box has at least two GPUs (the round's driver
  runs the multi-GPU bench on an 8-GPU node; this test lights up wherever 2 GPUs are visible).
The ranks are child processes started by torch.distributed.run (never more than 3 of them on the card)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


def _run(backend, world):
    port = 29700 + (os.getpid() % 1500) + 17 * world + (0 if backend == "gloo" else 5)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "mp_gpu_worker.py"), backend]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count(": ok") == 6 and "MISMATCH" not in res.stdout, res.stdout  # 3 scenes x {synchronous, pipelined}


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_over_gloo(world):
    _run("gloo", world)


def test_one_rank_over_rccl():
    """RCCL itself on this image and box: process-group set-up bound to the device, barrier, broadcast and the exchange's code
    path with a single rank (what a one-GPU box can say about the backend the multi-GPU runs use)."""
    _run("nccl", 1)


def test_two_ranks_over_rccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL over xGMI)")
    _run("nccl", 2)
