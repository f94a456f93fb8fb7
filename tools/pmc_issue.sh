#!/bin/bash
# Vector-ALU instruction issue of one frame of a bench workload (rocprofv3 PMC, SQ block only - its own pass).
# usage (GPU box, repo root):  bash tools/pmc_issue.sh cfg4   ->  gpurun_out/issue_cfg4.json  (copy to profiles/)
set -e
W=${1:-cfg4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/issue_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU -d $OUT/sq -o p --output-format csv -- python3 $R/bench.py --workload $W --steps 1 --warmup 1 --no-extra --no-cpu-baseline > $OUT/sq.json 2> $OUT/sq.err
python3 $R/tools/pmc_issue.py $W $OUT/sq/p_counter_collection.csv > $R/gpurun_out/issue_$W.json
cat $R/gpurun_out/issue_$W.json
