"""CPU: `bench.py --gpus N` without a launcher starts its own N ranks as CHILD processes (torch.distributed.run), relays
rank 0's line, returns the child's code - and the launching process never touches the GPU."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
FAKE = ROOT / "tests" / "fake_bench_rank.py"


def _bench():
    sys.path.insert(0, str(ROOT))
    import bench
    return bench


def test_launcher_spawns_n_children_and_relays_rank0(tmp_path, capsys, monkeypatch):
    bench = _bench()
    import torch
    monkeypatch.setenv("RT_BENCH_ONE_GPU", "1")
    monkeypatch.setenv("RT_TEST_MARKER_DIR", str(tmp_path))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = bench.launch_ranks(3, ["--gpus", "3", "--steps", "1"], script=FAKE)
    assert rc == 0
    assert sorted(p.name for p in tmp_path.iterdir()) == ["rank0", "rank1", "rank2"]  # three children ran
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["n_gpus"] == 3 and d["launcher"] == "self" and d["backend"] == "gloo" and d["master"] == "127.0.0.1"
    assert d["argv"] == ["--gpus", "3", "--steps", "1"]
    assert not torch.cuda.is_initialized()  # the parent made no GPU call


def test_launcher_refuses_more_ranks_than_gpus(capsys, monkeypatch):
    bench = _bench()
    import torch
    monkeypatch.delenv("RT_BENCH_ONE_GPU", raising=False)
    n = torch.cuda.device_count() + 1
    called = []
    rc = bench.launch_ranks(max(n, 2), [], script=FAKE, run=lambda *a, **k: called.append(a))
    assert rc == 2 and not called
    assert "RT_BENCH_ONE_GPU" in capsys.readouterr().err


def test_launcher_passes_the_childs_failure_on(monkeypatch):
    bench = _bench()
    monkeypatch.setenv("RT_BENCH_ONE_GPU", "1")

    class Res:
        returncode = 7
        stdout = ""
    assert bench.launch_ranks(2, [], script=FAKE, run=lambda *a, **k: Res()) == 7


def test_bench_main_goes_through_the_launcher_when_no_launcher_started_it(tmp_path):
    """`python bench.py --gpus 2` end to end: the parent must hand over to children before importing anything that
    initialises a GPU; with no GPU in this container and no RT_BENCH_ONE_GPU it refuses with exit code 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RT_BENCH_ONE_GPU")}
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU node would really start the bench")
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 2 and "--gpus 2" in res.stderr
