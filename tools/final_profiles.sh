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
RT_WALK_STATS=1 python3 bench.py --no-cpu-baseline --no-extra --steps 1 --warmup 0 2> $F/walk_stats_raw.txt > /dev/null; grep walk $F/walk_stats_raw.txt > $F/walk_stats.txt; rm -f $F/walk_stats_raw.txt
for w in cfg2 cfg3 cfg5base cfg5; do python3 bench.py --workload $w --no-cpu-baseline > $F/bench_$w.json 2> $F/bench_$w.err; echo "$w done"; done
python3 bench.py --workload cfg4 --ray-buffer --no-cpu-baseline --no-extra > $F/bench_cfg4_raybuffer.json 2> /dev/null || true
for n in 2 4; do RT_BENCH_ONE_GPU=1 python3 bench.py --gpus $n --no-cpu-baseline --no-extra > $F/bench_gloo_one_gpu_n$n.json 2> $F/bench_gloo_n$n.err || true; done   # (bench.py starts its own ranks)
rm -rf gpurun_out/frame_cfg4 gpurun_out/frame_cfg3
ls -la $F
