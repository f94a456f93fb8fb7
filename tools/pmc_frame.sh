#!/bin/bash
# One frame of a bench workload under rocprofv3, four separate passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE
# cannot share a pass; counters never together with the kernel trace):
#   1. --kernel-trace --stats          per-kernel durations
#   2. --pmc FETCH_SIZE                3. --pmc WRITE_SIZE
#   4. --pmc SQ_* (instruction issue)
# -> gpurun_out/frame_<workload>.json (+ the kernel-stats csv), stamped with the sha-256 of the library that ran.
# Copy both into profiles/; bench.py only quotes the counters when the stamp matches the library it is running.
# usage (GPU box, repo root):  bash tools/pmc_frame.sh cfg4 [--ray-buffer] [--literal] [--no-grid]
# (flags that change what the kernels do or read go into the file's name and its "flags" field: bench.py only quotes a
#  counter file measured with the flags of its own run)
set -e
W=${1:-cfg4}
shift || true
FLAGS="$*"
KEY=$W$(for f in $(echo $FLAGS | tr ' ' '\n' | sort); do printf "+%s" "${f#--}"; done)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/frame_$KEY
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the frame the counters are read from is the LAST one of the run - one of bench.py's synchronous Render() calls
# (render_wall_ms_incl_d2h); since round 4 rt_render cuts a large frame into two passes, so keep it to one here: a whole frame
export RT_RENDER_PASSES=1
BENCH="python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-extra --no-cpu-baseline $FLAGS"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o p --output-format csv -- $BENCH > $OUT/trace.json 2> $OUT/trace.err
echo "pass 1 (kernel trace) done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $OUT/$C -o p --output-format csv -- $BENCH > $OUT/$C.json 2> $OUT/$C.err
  echo "pass $C done"
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU -d $OUT/sq -o p --output-format csv -- $BENCH > $OUT/sq.json 2> $OUT/sq.err
echo "pass SQ done"
RT_PMC_FLAGS="$FLAGS" python3 $R/tools/pmc_frame.py $W $OUT > $R/gpurun_out/frame_$KEY.json
cp $(ls $OUT/trace/*kernel_stats.csv | head -1) $R/gpurun_out/frame_${KEY}_kernel_stats.csv
cat $R/gpurun_out/frame_$KEY.json
