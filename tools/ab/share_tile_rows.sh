#!/bin/bash
# rank 0's share of the cfg4 frame at world 1 / 2 / 4 / 8 for other tile heights (rows of the interleaved split)
for r in 16 32 64 128 256 512; do
  python3 tools/ab/share_time.py $r 2> /dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["tile_rows"], {k: round(v, 3) for k, v in d["share_ms"].items()}, {k: round(v, 2) for k, v in d["speedup_before_gather"].items()})'
done
