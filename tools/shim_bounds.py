#!/usr/bin/env python3
"""How far does the REFERENCE's output move when its OpenCL runtime defines dot() / normalize() differently?

The reference pins no OpenCL implementation; the oracle's shim (oracle/ref_shim.cl) chooses one conforming definition of the
two geometric builtins (plain left-to-right dot, normalize = v / sqrt). This script runs every shade / shade_and_reflect
golden fixture (plus larger synthetic frames) through the reference's verbatim kernels linked against three OTHER conforming
definitions (fma-chain dot; reciprocal-multiply normalize; both) and - round 4 - against AMD's OWN OpenCL builtin library as
restated from /opt/rocm/amdgcn/bitcode/opencl.bc (variants 4-6: fmuladd-chain dot, normalize = v * rsqrt(dot) with a zero
vector returned unchanged, v_rsq_f32 modelled as correctly rounded / one ulp up / one ulp down) and reports, against the default shim: max |dRGB| over
the pixels that stay within 1e-5 ("stable"), and the number of pixels that do not (a secondary ray flipped between hit and
miss at a silhouette / shadow edge - implementation-defined in the reference itself).

Build container only (needs oracle/_ref). Writes profiles/r04_shim_bounds.json (r02_shim_bounds.json = round 2's run with
variants 1-3 only) and tests/golden_alt/shim_variants.npz (the variant outputs of a subset of fixtures, which the GPU tests
hold the HIP backend against).
    python tools/shim_bounds.py
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import camera, fixture_names, load_fixture, random_scene
from opencl_raytracer_amd import synthetic
from oracle import oracle

KN = {1: "shade", 2: "shade_and_reflect"}
VARIANTS = {1: "dot = fma chain", 2: "normalize = v * (1/sqrt)", 3: "both",
            4: "AMD opencl.bc: dot = fmuladd chain, normalize = v * rsqrt(dot) (zero vector unchanged), rsqrt correctly rounded",
            5: "AMD opencl.bc, rsqrt one ulp up", 6: "AMD opencl.bc, rsqrt one ulp down"}
VS = tuple(sorted(VARIANTS))
SUBSET = ["scene_roundedCube_64_shade_and_reflect", "scene_simpleScene_64_shade_and_reflect", "scene_multipleSpheres_64_shade",
          "random_mixed100_shade_and_reflect", "random_mixed100_shade", "synthetic_1k_shade_and_reflect", "stale_specular_shade_and_reflect",
          "lights_012_shade", "directional_shade_and_reflect", "bounce_a0.2_D3", "corridor_D30", "shipped_roundedCube_160x90_D30"]


def run_all(kernel, objs, lights, rays, D):
    outs = [np.ascontiguousarray(oracle.Reference(kernel, True, v).render(objs, lights, rays, D)["out"][:, :3]) for v in (0,) + VS]
    return outs


def compare(outs):
    base = outs[0].astype(np.float64)
    res = {}
    for v in VS:
        d = np.abs(outs[v].astype(np.float64) - base)
        d = np.where(np.isnan(outs[v]) & np.isnan(outs[0]), 0.0, d)
        per_pixel = np.nanmax(d, axis=1) if len(d) else np.zeros(0)
        unstable = per_pixel > 1e-5
        res[v] = {"max_abs_stable": float(per_pixel[~unstable].max()) if (~unstable).any() else 0.0,
                  "unstable_pixels": int(unstable.sum()), "max_abs_unstable": float(per_pixel[unstable].max()) if unstable.any() else 0.0,
                  "mask_changes": int((np.any(outs[v] != 0, axis=1) != np.any(outs[0] != 0, axis=1)).sum())}
    return res


def main():
    if not oracle.reference_available():
        sys.exit("needs oracle/_ref (build container)")
    rows, store = [], {}
    for name in fixture_names():
        fx = load_fixture(name)
        if fx["kernel"] == 0 or name.startswith("degenerate"):
            continue
        outs = run_all(KN[fx["kernel"]], fx["objs"], fx["lights"], fx["rays"], fx["max_bounces"])
        assert np.array_equal(outs[0].view(np.uint32), fx["out_fused"].view(np.uint32)), name  # the default shim IS the golden vector
        rows.append({"case": name, "pixels": len(fx["rays"]), "hit_pixels": int(np.any(outs[0] != 0, axis=1).sum()), **{f"v{v}": r for v, r in compare(outs).items()}})
        if name in SUBSET:
            store[name] = np.stack(outs[1:])
    # larger frames: the config-4 generator at 10 000 spheres / 8 lights (96 x 64), a 200-object mixed scene, deep bounces
    extra = [("synthetic_10k_L8_96x64_D3", "shade_and_reflect", *synthetic.spheres_and_lights(10_000, 8), camera.primary_rays(96, 64), 3),
             ("random_mixed200_128x96_D5", "shade_and_reflect", *random_scene(120, 80, 4, seed=91, directional_lights=1, spread=9.0), camera.primary_rays(128, 96), 5),
             ("random_mixed200_128x96_shade", "shade", *random_scene(120, 80, 4, seed=91, directional_lights=1, spread=9.0), camera.primary_rays(128, 96), 0)]
    for name, kernel, objs, lights, rays, D in extra:
        outs = run_all(kernel, objs, lights, rays, D)
        rows.append({"case": name, "pixels": len(rays), "hit_pixels": int(np.any(outs[0] != 0, axis=1).sum()), **{f"v{v}": r for v, r in compare(outs).items()}})
    tot = {v: {"max_abs_stable": max(r[f"v{v}"]["max_abs_stable"] for r in rows), "unstable_pixels": sum(r[f"v{v}"]["unstable_pixels"] for r in rows),
               "mask_changes": sum(r[f"v{v}"]["mask_changes"] for r in rows)} for v in VS}
    # A light AT the hit point (fixtures nan_shadow_*): the shim's normalize(0) is NaN, so its shadow ray starts at NaN; AMD's
    # returns the zero vector, so the ray starts at the hit point itself with direction 0 - what do the kernels make of that?
    light_at_hit = {}
    for name in fixture_names():
        if not name.startswith("nan_shadow"):
            continue
        fx = load_fixture(name)
        outs = run_all(KN[fx["kernel"]], fx["objs"], fx["lights"], fx["rays"], fx["max_bounces"])
        light_at_hit[name] = {f"v{v}": {"pixels_differing_from_shim": int((np.abs(outs[v].astype(np.float64) - outs[0]).max(axis=1) > 0).sum()),
                                         "max_abs": float(np.nanmax(np.abs(outs[v].astype(np.float64) - outs[0])))} for v in VS}
    summary = {"what": "reference kernels (fused) under other conforming dot()/normalize() definitions vs the oracle's shim", "variants": VARIANTS,
               "light_at_the_hit_point": light_at_hit,
               "pixels": sum(r["pixels"] for r in rows), "hit_pixels": sum(r["hit_pixels"] for r in rows), "totals": tot, "rows": rows}
    with open(os.path.join(ROOT, "profiles", "r04_shim_bounds.json"), "w") as f:
        json.dump(summary, f, indent=1)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden_alt", "shim_variants.npz"), **store)
    print(json.dumps({"pixels": summary["pixels"], "hit_pixels": summary["hit_pixels"], "totals": tot}, indent=1))


if __name__ == "__main__":
    main()
