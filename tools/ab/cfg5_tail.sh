#!/bin/bash
# cfg5: the frame's tail - who is still alive after each round, and the frame time under other hand-over thresholds of wf_finish
RT_ROUND_STATS=1 python3 bench.py --workload cfg5 --no-cpu-baseline --no-extra --steps 1 --warmup 0 2>&1 | grep -E "rounds|ms_per_step" | cut -c1-200
for t in 524288 131072 32768 8192 1; do
  echo "threshold $t: $(RT_WF_FINISH_THRESHOLD=$t python3 bench.py --workload cfg5 --no-cpu-baseline --no-extra | python3 -c 'import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["ms_per_step"])')"
done
