#!/usr/bin/env python3
"""Rank 0's share of the cfg4 frame at world 4 / 8 for different wf_finish thresholds (RT_WF_FINISH_THRESHOLD)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for th in (None, 8192, 32768, 65536, 131072):
    env = dict(os.environ)
    if th: env["RT_WF_FINISH_THRESHOLD"] = str(th)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools/ab/share_time.py"), "16"], capture_output=True, text=True, env=env)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])["share_ms"]
        print(th, {k: round(v, 3) for k, v in d.items()}, flush=True)
    except Exception:
        print("FAILED", th, r.stderr[-300:], flush=True)
