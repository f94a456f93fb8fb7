#!/usr/bin/env python3
"""Writes scenes/*.txt - the inputs BASELINE.json's configs[0..2] and configs[4] name - from a structured description.

The four scenes hold the same content as the reference's sample scenes of the same names (they ARE the workloads the
metric is quoted on: same materials, transforms, primitives, light); the text is emitted here in the reference's scene
grammar (SceneLoader.cpp:50-331 parses it): header blocks `material` / `light` with 2-space
indented properties, a line `===`, then a body whose transforms scope their children by indentation. Nothing is copied
from the reference tree; `tests/test_host_logic.py` checks that the committed files are exactly what this prints.

    python scenes/make_scenes.py          # rewrite scenes/*.txt
"""
from __future__ import annotations

from pathlib import Path

HERE = Path(__file__).resolve().parent

LIGHTING_TEST = ("lightingTest", [("ambient", "1 0 0"), ("diffuse", "0 1 0"), ("specular", "0 0 1")])
GLOBAL_LIGHT = ("globalLight", [("ambient", ".3 .3 .3"), ("diffuse", ".7 .7 .7"), ("specular", "1 1 1")])


def sphere(mat="lightingTest"):
    return ("primative", "sphere", mat)   # (sic) the grammar's keyword


def box(mat="lightingTest"):
    return ("primative", "box", mat)


def light(name="globalLight"):
    return ("light", name)


def xf(kind, args, *children):
    return ("xf", kind, args, list(children))


def comment(text):
    return ("#", text)


def blank():
    return ("",)


SCENES = {
    "simpleSphere": dict(
        head=["BASELINE config 1: one unit sphere at the view-space point (0,0,-10), one positional light.",
              "Scene content equivalent to the reference's simpleSphere sample; grammar: SURVEY.md Appendix A."],
        materials=[LIGHTING_TEST], lights=[GLOBAL_LIGHT],
        body=[comment("body: objects/lights are emitted with the matrix on top of the stack"), sphere(),
              xf("translate", "10 10 10", light())]),
    "multipleSpheres": dict(
        head=["BASELINE config 2: three spheres (one non-uniformly scaled: kernel normals use mv, quirk Q2).",
              "Scene content equivalent to the reference's multipleSpheres sample."],
        materials=[LIGHTING_TEST], lights=[GLOBAL_LIGHT],
        body=[sphere(), comment("ellipsoid, upper right"),
              xf("translate", "4 2 0", xf("scale", "2 3 1", sphere())),
              comment("radius-2 sphere, lower left"),
              xf("translate", "-3 -4 0", xf("scale", "2 2 2", sphere())),
              xf("translate", "10 10 10", light())]),
    "simpleScene": dict(
        head=["BASELINE config 3: a rotated box and a sphere under a common scale; the light is a child of",
              "`scale 2 2 2`, so it sits at view-space (20,20,10).",
              "Scene content equivalent to the reference's simpleScene sample."],
        materials=[LIGHTING_TEST], lights=[GLOBAL_LIGHT],
        body=[blank(),
              xf("scale", "2 2 2",
                 xf("translate", "1 0 0", xf("rotate", "45 1 1 1", xf("scale", "2 2 2", box()))),
                 sphere(),
                 xf("translate", "10 10 10", light()))]),
    "roundedCube": dict(
        head=["BASELINE config 5 (analytic base): a mirror box with eight spheres on its corners.",
              "Two materials with absorption .7 / .2 -> real multi-bounce paths.",
              "Scene content equivalent to the reference's roundedCube sample."],
        materials=[("lightingTest", LIGHTING_TEST[1] + [("absorption", ".7"), ("reflection", ".3")]),
                   ("mirror", [("ambient", ".1 .1 .1"), ("diffuse", ".6 .6 .6"), ("specular", "1 1 1"), ("shininess", "100"),
                               ("absorption", ".2"), ("reflection", ".8")])],
        lights=[GLOBAL_LIGHT],
        body=[xf("rotate", "45 1 1 1",
                 xf("scale", "4 4 4", box("mirror")),
                 blank(),
                 comment("the eight corners (+-2, +-2, +-2)"),
                 *[xf("translate", f"{x} {y} {z}", sphere()) for y in (2, -2) for z in (2, -2) for x in (2, -2)]),
              blank(),
              xf("translate", "10 10 10", light())]),
}


def emit_body(nodes, depth, out):
    pad = "  " * depth
    for n in nodes:
        if n[0] == "":
            out.append("")
        elif n[0] == "#":
            out.append(f"{pad}# {n[1]}")
        elif n[0] == "primative":
            out.append(f"{pad}primative {n[1]} {n[2]}")
        elif n[0] == "light":
            out.append(f"{pad}light {n[1]}")
        else:
            _, kind, args, children = n
            out.append(f"{pad}{kind} {args}")
            emit_body(children, depth + 1, out)


def scene_text(name):
    d = SCENES[name]
    out = [f"# {line}" for line in d["head"]]
    for kind, blocks in (("material", d["materials"]), ("light", d["lights"])):
        for bname, props in blocks:
            out.append(f"{kind} {bname}")
            out += [f"  {k} {v}" for k, v in props]
            out.append("")
    out.append("===")
    emit_body(d["body"], 0, out)
    return "\n".join(out) + "\n"


def main():
    for name in SCENES:
        (HERE / f"{name}.txt").write_text(scene_text(name))
        print("wrote", HERE / f"{name}.txt")


if __name__ == "__main__":
    main()
