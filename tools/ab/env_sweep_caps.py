#!/usr/bin/env python3
"""Wave caps of a round's two persistent walks (RT_WAVES_CLOSEST / RT_WAVES_ANY, run-time knobs of launch_persistent) against the
cfg4 frame time; every combination twice, interleaved with the default. usage: python tools/ab/env_sweep_caps.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
combos = [{}] + [{"RT_WAVES_CLOSEST": c, "RT_WAVES_ANY": a} for c in ("4864", "5120", "5376") for a in ("2560", "3072", "3584", "4096")]
for rep in range(2):
    for env_add in combos:
        env = dict(os.environ); env.update(env_add)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--steps", "8", "--warmup", "2"] + sys.argv[1:],
                           capture_output=True, text=True, env=env)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{str(env_add):60s} {d['ms_per_step']:8.3f} ms", flush=True)
