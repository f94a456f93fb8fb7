"""Sum the SQ instruction counters over the kernels of the LAST frame of a bench run -> issue JSON.

SQ_INSTS_VALU counts wave-level vector instructions; SQ_ACTIVE_INST_VALU their issue time in quad-cycles (measured
here: 1.01-1.03 per instruction, i.e. ~4 cycles each for this instruction mix). Peak issue of the chip for such
instructions: 256 CU x 4 SIMD x 2.4 GHz / 4 cycles = 614 G wave-instructions/s."""
import collections, csv, json, sys

workload, path = sys.argv[1:3]
rows = [r for r in csv.DictReader(open(path)) if "rt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
disp = sorted({int(r["Dispatch_Id"]) for r in rows})
starts = [int(r["Dispatch_Id"]) for r in rows if "wf_begin" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU"]
first = starts[-1] if starts else disp[-1]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if int(r["Dispatch_Id"]) >= first:
        per[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
tot = collections.defaultdict(float)
for k in per.values():
    for c, v in k.items():
        tot[c] += v
print(json.dumps({
    "workload": f"{workload}: one frame, all kernels of the frame (last frame of `bench.py --workload {workload} --steps 1 --warmup 1`)",
    "valu_wave_instructions": tot["SQ_INSTS_VALU"], "valu_active_quad_cycles": tot["SQ_ACTIVE_INST_VALU"],
    "salu_wave_instructions": tot["SQ_INSTS_SALU"], "wave_quad_cycles": tot["SQ_WAVE_CYCLES"], "wait_quad_cycles": tot["SQ_WAIT_ANY"],
    "lanes_per_valu_instruction": tot["SQ_THREAD_CYCLES_VALU"] / max(tot["SQ_ACTIVE_INST_VALU"], 1.0),
    "per_kernel": {k: dict(v) for k, v in per.items()},
    "peak_valu_wave_instructions_per_s": 256 * 4 * 2.4e9 / 4,
    "source": "rocprofv3 --pmc SQ_* (tools/pmc_issue.sh)",
}, indent=1))
