#!/usr/bin/env python3
"""Rank 0's share of the cfg4 frame at world 8 (and 4) under different wave caps of the two walks (RT_WAVES_CLOSEST / RT_WAVES_ANY).
usage: python tools/ab/share8_caps.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for cc, ca in ((None, None), (4096, 2048), (3072, 2048), (4096, 1024), (3072, 1024), (2048, 2048), (6144, 2048), (4096, 4096)):
    env = dict(os.environ)
    if cc: env["RT_WAVES_CLOSEST"] = str(cc); env["RT_WAVES_ANY"] = str(ca)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools/ab/share_time.py"), "16"], capture_output=True, text=True, env=env)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])["share_ms"]
        print(cc, ca, {k: round(v, 3) for k, v in d.items()}, flush=True)
    except Exception:
        print("FAILED", cc, ca, r.stderr[-300:], flush=True)
