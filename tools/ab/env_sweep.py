#!/usr/bin/env python3
"""Frame time of bench.py's workload under run-time tuning knobs (environment variables the library reads per launch / at
rt_create), one child process each. usage: python tools/ab/env_sweep.py [bench.py args ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SWEEP = [{}] + [{"RT_WAVES_ANY": v} for v in ("1024", "2048", "3072", "6144", "8192")] + \
        [{"RT_WAVES_CLOSEST": v} for v in ("4096", "5120", "6144", "7168")] + \
        [{"RT_WALK_BLOCK_FACTOR": v} for v in ("1.5", "1.6", "1.75", "1.9")] + \
        [{"RT_LT_TILE_FACTOR": v} for v in ("1.2", "2.0", "2.5")] + \
        [{"RT_WF_FINISH_THRESHOLD": v} for v in ("32768", "65536", "262144")] + \
        [{"RT_GRID_CELLS_PER_OBJECT": v} for v in ("2", "4")] + [{}]
if os.environ.get("SWEEP_VAR"):  # SWEEP_VAR=NAME SWEEP_VALUES=a,b,c: one knob, the default before and after
    SWEEP = [{}] + [{os.environ["SWEEP_VAR"]: v} for v in os.environ.get("SWEEP_VALUES", "").split(",") if v] + [{}]
for env_add in SWEEP:
    env = dict(os.environ); env.update(env_add)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "5", "--warmup", "2"] + sys.argv[1:],
                       capture_output=True, text=True, env=env)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{str(env_add):44s} {d['ms_per_step']:8.3f} ms", flush=True)
    except Exception:
        print("FAILED", env_add, r.stderr[-300:], flush=True)
