#!/bin/bash
# The round's evidence in one go (GPU box, repo root): counters + kernel trace of one cfg4 / cfg3 frame (stamped with the
# library hash), then the default bench line (which quotes those counters), its kernel stats, the multi-GPU share
# rehearsal, the other workloads. Everything lands in gpurun_out/final/; copy what is to be judged into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
F=$R/gpurun_out/final
mkdir -p $F
cd $R
bash tools/pmc_frame.sh cfg4 > $F/pmc_cfg4.log 2>&1
bash tools/pmc_frame.sh cfg3 > $F/pmc_cfg3.log 2>&1
cp gpurun_out/frame_cfg4.json gpurun_out/frame_cfg3.json profiles/   # on the box: bench.py reads profiles/
cp gpurun_out/frame_cfg4.json gpurun_out/frame_cfg3.json gpurun_out/frame_cfg4_kernel_stats.csv gpurun_out/frame_cfg3_kernel_stats.csv $F/
echo "counters done"
python3 bench.py > $F/bench.json 2> $F/bench.err
echo "bench done"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $F/default_trace -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $F/bench_traced.json 2> $F/bench_traced.err)
cp $(ls $F/default_trace/*kernel_stats.csv | head -1) $F/default_bench_kernel_stats.csv
rm -rf $F/default_trace
echo "trace done"
./tools/ubench/issue_forms > $F/issue_forms.txt 2>&1 || true
python3 tools/ab/share_time.py 16 > $F/share_rehearsal.json 2> /dev/null
RT_WALK_STATS=1 python3 bench.py --no-cpu-baseline --no-extra --steps 1 --warmup 0 2> $F/walk_stats_raw.txt > /dev/null; grep -E "walk|blocks|grid" $F/walk_stats_raw.txt > $F/walk_stats.txt; rm -f $F/walk_stats_raw.txt
for w in cfg2 cfg3 cfg5base cfg5; do python3 bench.py --workload $w --no-cpu-baseline > $F/bench_$w.json 2> $F/bench_$w.err; echo "$w done"; done
python3 bench.py --workload cfg4 --ray-buffer --no-cpu-baseline --no-extra > $F/bench_cfg4_raybuffer.json 2> /dev/null || true
for n in 2 4; do RT_BENCH_ONE_GPU=1 python3 bench.py --gpus $n --no-cpu-baseline --no-extra > $F/bench_gloo_one_gpu_n$n.json 2> $F/bench_gloo_n$n.err || true; done   # (bench.py starts its own ranks)
# kernel timelines of one frame (RT_RENDER_PASSES=1: the last frame of a run is one of bench.py's Render() calls - keep it whole):
export RT_RENDER_PASSES=1
# ... the whole cfg4 frame, the same with the two walks of a round one after the other, rank 0's share at world 8, cfg5
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $F/tr_a -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2>&1) || true
python3 tools/ab/timeline.py $(find $F/tr_a -name "*kernel_trace.csv" | head -1) > $F/timeline_cfg4.txt 2>&1 || true
(cd /tmp && export TMPDIR=/tmp && RT_WF_ONE_STREAM=1 rocprofv3 --kernel-trace -d $F/tr_b -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2>&1) || true
python3 tools/ab/timeline.py $(find $F/tr_b -name "*kernel_trace.csv" | head -1) > $F/timeline_cfg4_one_stream.txt 2>&1 || true
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $F/tr_c -o p --output-format csv -- python3 $R/tools/ab/share_trace.py 8 > /dev/null 2>&1) || true
python3 tools/ab/timeline.py $(find $F/tr_c -name "*kernel_trace.csv" | head -1) > $F/timeline_cfg4_share_of_8.txt 2>&1 || true
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $F/tr_d -o p --output-format csv -- python3 $R/bench.py --workload cfg5 --steps 2 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2>&1) || true
python3 tools/ab/timeline.py $(find $F/tr_d -name "*kernel_trace.csv" | head -1) > $F/timeline_cfg5.txt 2>&1 || true
rm -rf $F/tr_a $F/tr_b $F/tr_c $F/tr_d gpurun_out/frame_cfg4 gpurun_out/frame_cfg3
ls -la $F
