#!/usr/bin/env python3
"""How close to issue-bound is the closest-hit block walk? (replace "97 % busy by SQ_ACTIVE_INST_VALU" - a counter
that restates the instruction count at an assumed 4 cycles each - with arithmetic.)

Takes the ISA of rt::wf_walk_blocks<true, false, false> (hipcc -S), classifies the vector instructions of its persistent loop
by ISSUE FORM, prices every form with the cycles per wave-instruction tools/ubench/issue_forms.hip measured on this chip at 4-7
waves per SIMD (profiles/r03_issue_forms.txt), and compares  (vector instructions per wave-trip by PMC) x (mean price of the
static mix)  with the SIMD-cycles a wave-trip really takes  (launch time x SIMDs x clock / wave-trips).  The static mix of the loop
stands in for the dynamic one (hand-out and exact-test code run less often than the trip proper: stated, not hidden).

usage (build container): python tools/walk_trip_histogram.py <frame_cfg4.json> <walk_stats.txt>  > profiles/rNN_walk_trip_histogram.txt"""
import json, re, subprocess, sys, tempfile, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN2rt14wf_walk_blocksILb1ELb0ELb0EEEvNS_8WfParamsEPj"
# cycles per wave-instruction at 4-7 waves per SIMD (profiles/r03_issue_forms.txt; transcendental = quarter rate by the ISA guide)
PRICE = {"plain (all-VGPR fma / mul / add / cvt / logic)": 2.9, "SGPR or literal-with-SGPR source": 4.4, "v_cmp": 5.0,
         "v_cndmask": 5.3, "min3 / max3 / med3": 5.3, "transcendental (rcp / rsq / sqrt / exp / log)": 11.6,
         "cross-lane (readlane / readfirstlane / mbcnt / permute)": 4.4, "vector memory (global / scratch / flat)": 4.0, "LDS": 4.0}

def classify(line):
    op = line.split()[0]
    args = line[len(op):]
    if op.startswith(("global_", "scratch_", "flat_", "buffer_")): return "vector memory (global / scratch / flat)"
    if op.startswith("ds_"): return "LDS"
    if not op.startswith("v_"): return None
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_mbcnt", "v_permlane")): return "cross-lane (readlane / readfirstlane / mbcnt / permute)"
    if op.startswith("v_cmp"): return "v_cmp"
    if op.startswith("v_cndmask"): return "v_cndmask"
    if re.match(r"v_(min3|max3|med3)", op): return "min3 / max3 / med3"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", op): return "transcendental (rcp / rsq / sqrt / exp / log)"
    srcs = args.split(",")[1:]
    if any(re.match(r"\s*(s\d+|s\[\d+:\d+\]|vcc|exec)", a) for a in srcs): return "SGPR or literal-with-SGPR source"
    return "plain (all-VGPR fma / mul / add / cvt / logic)"

def main():
    prof = json.load(open(sys.argv[1]))
    stats = open(sys.argv[2]).read()
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "w.s")
        src = os.path.join(ROOT, "opencl-raytracer_amd", "csrc")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                        "-fno-fast-math", f"-I{ROOT}/include", f"-I{src}", "-S", "--cuda-device-only", os.path.join(src, "rt_wavefront.hip"), "-o", asm],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    body = text[text.index(KERNEL + ":"):]
    body = body[:body.index("s_endpgm")]
    lines = [l.strip() for l in body.splitlines()]
    # the persistent loop: everything from the first block labelled "Loop Header" to the last line that says "in Loop" / "Parent Loop"
    idx = [i for i, l in enumerate(lines) if "Loop" in l and l.startswith(".LBB")]
    loop = lines[idx[0]:idx[-1] + 60] if idx else lines
    hist, salu = {}, 0
    for l in loop:
        if not l or l.startswith((";", ".", "//")): continue
        c = classify(l)
        if c: hist[c] = hist.get(c, 0) + 1
        elif l.startswith("s_") and not l.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch")): salu += 1
    n = sum(hist.values())
    mean = sum(PRICE[k] * v for k, v in hist.items()) / n
    k = next(v for name, v in prof["kernels"].items() if "wf_walk_blocks" in name)
    m = re.search(r"\[walk closest\] rays (\d+) wave trips (\d+)", stats)
    trips = int(m.group(2))
    valu_per_trip = k["valu_wave_instructions"] / trips
    simd_cycles_per_trip = k["ms"] * 1e-3 * 2.4e9 * 1024 / trips
    print(f"rt::wf_walk_blocks<true, false, false>: vector instructions of the persistent loop by issue form (static, {n} instructions; {salu} scalar ALU beside them)")
    for name, v in sorted(hist.items(), key=lambda kv: -kv[1]):
        print(f"  {v:5d}  {100.0 * v / n:5.1f} %  x {PRICE[name]:4.1f} cycles   {name}")
    print(f"mean price of the mix: {mean:.2f} cycles per wave-instruction (profiles/r03_issue_forms.txt, 4-7 waves per SIMD)")
    print(f"one cfg4 frame ({sys.argv[1]}, library {prof.get('lib_sha16')}): {k['launches']} launches, {k['ms']:.3f} ms, {k['valu_wave_instructions'] / 1e9:.3f} G vector wave-instructions, "
          f"{trips} wave-trips -> {valu_per_trip:.0f} vector instructions and {simd_cycles_per_trip:.0f} SIMD-cycles per wave-trip (256 CUs x 4 SIMDs x 2.4 GHz)")
    print(f"priced: {valu_per_trip:.0f} x {mean:.2f} = {valu_per_trip * mean:.0f} cycles of issue per wave-trip = {100.0 * valu_per_trip * mean / simd_cycles_per_trip:.0f} % of the {simd_cycles_per_trip:.0f} it takes "
          f"(at the guide's 2 cycles: {100.0 * valu_per_trip * 2 / simd_cycles_per_trip:.0f} %; at SQ_ACTIVE_INST_VALU's 4: {100.0 * valu_per_trip * 4 / simd_cycles_per_trip:.0f} %)")
    print(f"lanes per vector instruction over the launches: {k['lanes_per_valu_instruction']:.1f} of 64")

if __name__ == "__main__":
    main()
