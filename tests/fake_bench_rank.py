"""Stand-in for one bench.py rank (tests/test_bench_launcher.py): reads the launcher's environment, touches no GPU,
rank 0 prints a result line that says how many ranks were started."""
import json
import os
import sys

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
marker = os.environ.get("RT_TEST_MARKER_DIR")
if marker:
    open(os.path.join(marker, f"rank{rank}"), "w").write(" ".join(sys.argv[1:]))
if rank == 0:
    print("noise before the line")
    print(json.dumps({"metric": "fake", "n_gpus": world, "launcher": os.environ.get("RT_BENCH_LAUNCHER"),
                      "backend": os.environ.get("RT_BENCH_BACKEND"), "master": os.environ.get("MASTER_ADDR"), "argv": sys.argv[1:]}))
