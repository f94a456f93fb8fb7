#!/bin/bash
# HBM traffic of one frame of a bench workload from the rocprofv3 PMC counters, collected as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate passes.
# usage (on the GPU box, from the repo root):  bash tools/pmc_traffic.sh cfg4   ->  profiles/traffic_cfg4.json
set -e
W=${1:-cfg4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/traffic_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $OUT/$C -o p --output-format csv -- python3 $R/bench.py --workload $W --steps 1 --warmup 1 --no-extra --no-cpu-baseline > $OUT/$C.json 2> $OUT/$C.err
done
python3 $R/tools/pmc_traffic.py $W $OUT/FETCH_SIZE/p_counter_collection.csv $OUT/WRITE_SIZE/p_counter_collection.csv > $R/gpurun_out/traffic_$W.json
cat $R/gpurun_out/traffic_$W.json
