#!/usr/bin/env python3
"""Per-kernel timeline of the LAST frame in a rocprofv3 kernel trace csv. usage: timeline.py <p_kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "wf_begin" in r["Kernel_Name"] or "wf_identity_round" in r["Kernel_Name"]]
fr = rows[starts[-1]:] if starts else rows[-1:]
t0 = int(fr[0]["Start_Timestamp"])
for r in fr:
    n = r["Kernel_Name"].split("(")[0].replace("void rt::", "").replace("rt::", "")
    b, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{b / 1e6:8.3f} {e / 1e6:8.3f} {(e - b) / 1e6:7.3f}  {n[:58]:58s} grid {r['Grid_Size_X']}")
