#!/usr/bin/env python3
"""Rank 0's share of the cfg4 frame at world 8 (and 4) under different wave caps of the two walks (RT_WAVES_CLOSEST / RT_WAVES_ANY).
usage: python tools/ab/share8_caps.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for cc, ca in ((None, None), (4096, 2048), (5120, 2048), (5120, 1536), (4096, 1536), (3072, 1536), (5120, 2560), (4096, 2560), (3584, 2048), (4608, 2048)):
    env = dict(os.environ)
    if cc: env["RT_WAVES_CLOSEST"] = str(cc); env["RT_WAVES_ANY"] = str(ca)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools/ab/share_time.py"), "16"], capture_output=True, text=True, env=env)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])["share_ms"]
        print(cc, ca, {k: round(v, 3) for k, v in d.items()}, flush=True)
    except Exception:
        print("FAILED", cc, ca, r.stderr[-300:], flush=True)
