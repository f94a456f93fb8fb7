#!/bin/bash
# The end-game of the closest-hit walk: a build with -DRT_STATS_TAIL=1 reports, in the counted pass, the longest time a wave kept
# walking after the queue had run dry for it (its rays in flight drain without refill) and the longest ray of the frame in trips.
# usage (GPU box): bash tools/ab/tail_stats.sh     (builds the variant when it is not there)
set -e
cd "$(dirname "$0")/../.."
V=opencl-raytracer_amd/csrc/variants/libhip_raytracer_tailstats.so
[ -f $V ] || bash tools/ab/build_variant.sh tailstats "-DRT_STATS_TAIL=1" > /dev/null 2>&1
RT_LIB_OVERRIDE=$PWD/$V RT_WALK_STATS=1 python3 bench.py --no-cpu-baseline --no-extra --steps 1 --warmup 0 2>&1 | grep "walk closest" | python3 -c '
import re, sys
line = sys.stdin.read()
v = int(re.search(r"exact rounds (\d+)", line).group(1)); longest = int(re.search(r"hand-out rounds (\d+)", line).group(1))
print(f"longest end-game of a wave: {(v >> 32) / 100:.1f} us, {v & 0xffffffff} trips in it (instrumented build); longest ray of the frame: {longest} trips")'
