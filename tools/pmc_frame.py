"""Combine the four rocprofv3 passes of tools/pmc_frame.sh into one JSON: per kernel of ONE frame (the last frame of the
run) - launches, device time, HBM-side bytes (FETCH_SIZE / WRITE_SIZE), vector / scalar instruction issue.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts a 128-byte request as 64 bytes for wide
(16 B / lane) coalesced reads -> x2; for the 4..16-byte scattered gathers of the large-scene path the factor is
uncalibrated, so both the raw (x1) and the corrected (x2) figure are kept and the x2 one is the upper bound quoted.
WRITE_SIZE is exact. SQ_INSTS_VALU counts wave-level vector instructions, SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES /
SQ_WAIT_ANY are in quad-cycles (4 shader cycles)."""
import collections, csv, glob, hashlib, json, os, sys

workload, out_dir = sys.argv[1:3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.environ.get("RT_LIB_OVERRIDE", os.path.join(root, "opencl-raytracer_amd", "csrc", "libhip_raytracer.so"))
lib_sha16 = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def find(sub, pattern):
    hits = sorted(glob.glob(os.path.join(out_dir, sub, "**", pattern), recursive=True))
    if not hits:
        raise SystemExit(f"no {pattern} under {out_dir}/{sub}")
    return hits[0]


def last_frame_rows(rows, key_id):
    """rows of our kernels from the last frame start on (large scenes: wf_identity_round, or wf_begin for frames with
    padding work-items) or the last launch (small scenes)"""
    ours = [r for r in rows if "rt::" in r["Kernel_Name"]]
    ours.sort(key=lambda r: int(r[key_id]))
    starts = [i for i, r in enumerate(ours) if "wf_begin" in r["Kernel_Name"] or "wf_identity_round" in r["Kernel_Name"]]
    if starts:
        first_id = int(ours[starts[-1]][key_id])
        return [r for r in ours if int(r[key_id]) >= first_id]
    last_id = int(ours[-1][key_id])
    return [r for r in ours if int(r[key_id]) == last_id]


per = collections.defaultdict(lambda: collections.defaultdict(float))

# pass 1: kernel trace -> launches and device time of the last frame
trace = list(csv.DictReader(open(find("trace", "*kernel_trace.csv"))))
t0 = t1 = None
for r in last_frame_rows(trace, "Dispatch_Id"):
    k = short(r["Kernel_Name"])
    b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    per[k]["launches"] += 1
    per[k]["ns"] += e - b
    t0 = b if t0 is None else min(t0, b)
    t1 = e if t1 is None else max(t1, e)

# passes 2-4: counters
for sub, names in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]),
                   ("sq", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_THREAD_CYCLES_VALU"])):
    rows = list(csv.DictReader(open(find(sub, "*counter_collection.csv"))))
    for name in names:
        sel = [r for r in rows if r["Counter_Name"] == name]
        for r in last_frame_rows(sel, "Dispatch_Id"):
            per[short(r["Kernel_Name"])][name] += float(r["Counter_Value"])

HBM_PEAK = 8.0e12
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 2   # MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD-32
kernels = {}
tot = collections.defaultdict(float)
for k, v in per.items():
    ms = v["ns"] / 1e6
    fetch_raw, write = v["FETCH_SIZE"] * 1024.0, v["WRITE_SIZE"] * 1024.0
    traffic = 2.0 * fetch_raw + write
    kernels[k] = {
        "launches": int(v["launches"]), "ms": ms,
        "fetch_bytes_raw": fetch_raw, "write_bytes": write, "traffic_bytes": traffic,
        "traffic_GBps": traffic / (ms * 1e-3) / 1e9 if ms > 0 else None,
        "traffic_frac_of_hbm_peak": traffic / (ms * 1e-3) / HBM_PEAK if ms > 0 else None,
        "valu_wave_instructions": v["SQ_INSTS_VALU"], "salu_wave_instructions": v["SQ_INSTS_SALU"],
        "valu_active_quad_cycles": v["SQ_ACTIVE_INST_VALU"], "wave_quad_cycles": v["SQ_WAVE_CYCLES"], "wait_quad_cycles": v["SQ_WAIT_ANY"],
        "lanes_per_valu_instruction": v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"] if v["SQ_ACTIVE_INST_VALU"] else None,
        "valu_issue_frac": v["SQ_INSTS_VALU"] / (ms * 1e-3) / VALU_PEAK_WAVE_INSTR if ms > 0 else None,
    }
    for c in ("ns", "FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_THREAD_CYCLES_VALU"):
        tot[c] += v[c]
span_ms = (t1 - t0) / 1e6 if t0 is not None else None
traffic_total = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
print(json.dumps({
    "workload": f"{workload}: ONE frame (the last of `bench.py --workload {workload} --steps 2 --warmup 1`), every kernel of the frame",
    "lib_sha16": lib_sha16,
    "flags": sorted(f.lstrip("-") for f in os.environ.get("RT_PMC_FLAGS", "").split()),
    "frame_span_ms": span_ms, "kernel_ms_sum": tot["ns"] / 1e6,
    "hbm_bytes_per_launch": traffic_total,
    "fetch_bytes_raw": tot["FETCH_SIZE"] * 1024.0, "write_bytes": tot["WRITE_SIZE"] * 1024.0,
    "correction": "traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies 128-B read requests at 64 B; upper bound for narrow gathers)",
    "valu_wave_instructions": tot["SQ_INSTS_VALU"], "salu_wave_instructions": tot["SQ_INSTS_SALU"],
    "valu_active_quad_cycles": tot["SQ_ACTIVE_INST_VALU"], "wave_quad_cycles": tot["SQ_WAVE_CYCLES"], "wait_quad_cycles": tot["SQ_WAIT_ANY"],
    "lanes_per_valu_instruction": tot["SQ_THREAD_CYCLES_VALU"] / tot["SQ_ACTIVE_INST_VALU"] if tot["SQ_ACTIVE_INST_VALU"] else None,
    "peak_valu_wave_instructions_per_s": VALU_PEAK_WAVE_INSTR,
    "peak_note": "256 CU x 4 SIMD x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md, cycle constants; 4 cycles for one wave alone)",
    "kernels": kernels,
    "source": "tools/pmc_frame.sh: rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; --pmc SQ_* (four separate passes)",
}, indent=1))
