#!/bin/bash
# rank 0's share of the cfg4 frame at world 1 / 2 / 4 / 8 under different hand-over thresholds of wf_finish (absolute pixel counts;
# the default is max(2048, n / 128)). usage: bash tools/ab/finish_threshold.sh > out.txt
for t in default 24000 50000 65536 100000 200000 262144 400000; do
  if [ $t = default ]; then unset RT_WF_FINISH_THRESHOLD; else export RT_WF_FINISH_THRESHOLD=$t; fi
  echo -n "$t: "; timeout -k 10 120 python3 tools/ab/share_time.py 16 2> /dev/null | python3 -c 'import json,sys; print({k: round(v, 3) for k, v in json.loads(sys.stdin.read())["share_ms"].items()})'
done
