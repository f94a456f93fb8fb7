#!/bin/bash
# gpurun_out/final/ (tools/final_profiles.sh) -> profiles/ under the names profiles/README.md lists
# usage: tools/copy_profiles.sh [round tag, default r04]
set -e
cd "$(dirname "$0")/.."
F=gpurun_out/final
R=${1:-r04}
cp $F/frame_cfg4.json $F/frame_cfg3.json profiles/
cp $F/frame_cfg4_kernel_stats.csv profiles/${R}_frame_cfg4_kernel_stats.csv
cp $F/frame_cfg3_kernel_stats.csv profiles/${R}_frame_cfg3_kernel_stats.csv
cp $F/bench.json profiles/${R}_bench.json
cp $F/default_bench_kernel_stats.csv profiles/${R}_default_bench_kernel_stats.csv
cp $F/walk_stats.txt profiles/${R}_walk_stats.txt
cp $F/share_rehearsal.json profiles/${R}_share_rehearsal.json
for w in cfg2 cfg3 cfg5base cfg5; do cp $F/bench_$w.json profiles/${R}_bench_$w.json; done
for n in 2 4; do cp $F/bench_gloo_one_gpu_n$n.json profiles/${R}_bench_gloo_one_gpu_n$n.json; done
cp $F/issue_forms.txt profiles/${R}_issue_forms.txt
cp $F/bench_cfg4_raybuffer.json profiles/${R}_bench_cfg4_raybuffer.json
for t in timeline_cfg4 timeline_cfg4_one_stream timeline_cfg4_share_of_8 timeline_cfg5; do cp $F/$t.txt profiles/${R}_$t.txt; done
for f in multi_wall.json share_rehearsal_frame_kernel.json share_rehearsal_step_rounds.json bench_frame_kernel.json bench_step_rounds.json walk_stats_frame_kernel.txt timeline_cfg4_step_rounds.txt timeline_cfg4_frame_kernel.txt; do [ -f $F/$f ] && cp $F/$f profiles/${R}_$f || true; done
