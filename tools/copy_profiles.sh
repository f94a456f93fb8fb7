#!/bin/bash
# gpurun_out/final/ (tools/final_profiles.sh) -> profiles/ under the names profiles/README.md lists
set -e
cd "$(dirname "$0")/.."
F=gpurun_out/final
cp $F/frame_cfg4.json $F/frame_cfg3.json profiles/
cp $F/frame_cfg4_kernel_stats.csv profiles/r03_frame_cfg4_kernel_stats.csv
cp $F/frame_cfg3_kernel_stats.csv profiles/r03_frame_cfg3_kernel_stats.csv
cp $F/bench.json profiles/r03_bench.json
cp $F/default_bench_kernel_stats.csv profiles/r03_default_bench_kernel_stats.csv
cp $F/walk_stats.txt profiles/r03_walk_stats.txt
cp $F/share_rehearsal.json profiles/r03_share_rehearsal.json
for w in cfg2 cfg3 cfg5base cfg5; do cp $F/bench_$w.json profiles/r03_bench_$w.json; done
for n in 2 4; do cp $F/bench_gloo_one_gpu_n$n.json profiles/r03_bench_gloo_one_gpu_n$n.json; done
cp $F/issue_forms.txt profiles/r03_issue_forms.txt
cp $F/bench_cfg4_raybuffer.json profiles/r03_bench_cfg4_raybuffer.json
for t in timeline_cfg4 timeline_cfg4_one_stream timeline_cfg4_share_of_8 timeline_cfg5; do cp $F/$t.txt profiles/r03_$t.txt; done
