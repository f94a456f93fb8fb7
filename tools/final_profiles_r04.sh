#!/bin/bash
# Round 4's additional evidence (GPU box, repo root; a second call because one gpurun call is limited to 20 minutes): the
# boundary's wall clock through one / several contexts, and the two other frame organisations (opt-in, measured and kept out of
# the default path). Everything lands in gpurun_out/final/ next to tools/final_profiles.sh's files.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
F=$R/gpurun_out/final
mkdir -p $F
cd $R
# round 4: the boundary's wall clock through one / several contexts, and the two other frame organisations (opt-in, measured)
python3 tools/ab/multi_wall.py cfg4 cfg3 > $F/multi_wall.json 2> /dev/null || true
RT_FRAME_KERNEL=1 python3 bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 2 > $F/bench_frame_kernel.json 2> /dev/null || true
RT_STEP_ROUNDS=1 python3 bench.py --no-cpu-baseline --no-extra --steps 5 --warmup 2 > $F/bench_step_rounds.json 2> /dev/null || true
RT_FRAME_KERNEL=1 python3 tools/ab/share_time.py 16 > $F/share_rehearsal_frame_kernel.json 2> /dev/null || true
RT_STEP_ROUNDS=1 python3 tools/ab/share_time.py 16 > $F/share_rehearsal_step_rounds.json 2> /dev/null || true
RT_FRAME_KERNEL=1 RT_WALK_STATS=1 python3 bench.py --no-cpu-baseline --no-extra --steps 1 --warmup 0 2> $F/ws_raw.txt > /dev/null; grep -E "walk|blocks" $F/ws_raw.txt > $F/walk_stats_frame_kernel.txt; rm -f $F/ws_raw.txt
export RT_RENDER_PASSES=1   # (timelines: the last frame of a run is one of bench.py's Render() calls - keep it whole)
(cd /tmp && export TMPDIR=/tmp && RT_STEP_ROUNDS=1 rocprofv3 --kernel-trace -d $F/tr_e -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2>&1) || true
python3 tools/ab/timeline.py $(find $F/tr_e -name "*kernel_trace.csv" | head -1) > $F/timeline_cfg4_step_rounds.txt 2>&1 || true
(cd /tmp && export TMPDIR=/tmp && RT_FRAME_KERNEL=1 rocprofv3 --kernel-trace -d $F/tr_f -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2>&1) || true
python3 tools/ab/timeline.py $(find $F/tr_f -name "*kernel_trace.csv" | head -1) > $F/timeline_cfg4_frame_kernel.txt 2>&1 || true
rm -rf $F/tr_e $F/tr_f
ls -la $F
