#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the REFERENCE ITSELF.

Runs only in the build container: it needs oracle/_ref, i.e. the reference's OpenCL-C kernels
compiled verbatim for the host from /root/reference (oracle/Makefile). The reference ships no
tests or golden vectors of its own, so these fixtures - inputs plus the
reference kernels' outputs - are what pins parity on the GPU box, where /root/reference does not
exist. Fixture categories follow SURVEY.md section 4's recommended list.

Each fixture is one compressed .npz:
    objs, lights      raw record bytes (uint8) in the reference's device layouts
    rays              raw Ray bytes, or absent when `camera` = (W, H, fov_degrees) describes a pinhole grid
    kernel            0 hittest / 1 shade / 2 shade_and_reflect
    max_bounces
    out_fused         reference output, contraction on  (-ffp-contract=on -mfma)
    out_unfused       reference output, contraction off (-ffp-contract=off)
Outputs are the kernel's output buffer as the reference host would see it: pixels start as
{0,0,0,1} (OpenCLRaytracer.cpp:32) / hittest slots as MAX_FLOAT and are only written on a hit;
only RGB (or t) is stored.

Usage:  python tests/golden/make_golden.py        (rewrites every fixture; deterministic)
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from helpers import F, R, SCENES, camera, instance, random_scene, rotation  # noqa: E402
from opencl_raytracer_amd import scene_loader, synthetic  # noqa: E402
from oracle import oracle  # noqa: E402

KID = {"hittest": 0, "shade": 1, "shade_and_reflect": 2}


def reference_outputs(kernel, objs, lights, rays, D):
    outs = {}
    for flavour, fused in (("fused", True), ("unfused", False)):
        out = oracle.Reference(kernel, fused).render(objs, lights, rays, D)["out"]
        outs[flavour] = out if KID[kernel] == 0 else np.ascontiguousarray(out[:, :3])
    return outs


def save(name, kernel, objs, lights, D, rays=None, cam=None):
    assert (rays is None) != (cam is None)
    r = rays if rays is not None else camera.primary_rays(cam[0], cam[1], cam[2])
    outs = reference_outputs(kernel, objs, lights, r, D)
    payload = dict(objs=np.frombuffer(objs.tobytes(), dtype=np.uint8), lights=np.frombuffer(lights.tobytes(), dtype=np.uint8),
                   kernel=np.int32(KID[kernel]), max_bounces=np.uint32(D), out_fused=outs["fused"], out_unfused=outs["unfused"])
    if rays is not None:
        payload["rays"] = np.frombuffer(rays.tobytes(), dtype=np.uint8)
    else:
        payload["camera"] = np.array(cam, dtype=np.float64)
    path = HERE / f"{name}.npz"
    np.savez_compressed(path, **payload)
    nz = int((outs["fused"].reshape(len(r), -1)[:, 0] != (oracle.MAX_FLOAT if KID[kernel] == 0 else 0)).sum())
    print(f"{name:44s} {kernel:18s} D={D} rays={len(r):6d} N={len(objs):5d} L={len(lights):2d} "
          f"written={nz:6d} {path.stat().st_size / 1024:7.1f} KiB")


def mat(ambient=(0, 0, 0), diffuse=(0, 0, 0), specular=(0, 0, 0), absorption=1.0, shininess=1.0):
    return R.Material(ambient, diffuse, specular, absorption, 1.0 - absorption, 0.0, shininess)


def light(pos, ambient=(.3, .3, .3), diffuse=(.7, .7, .7), specular=(1, 1, 1)):
    return R.make_light(R.LightProperties(ambient, diffuse, specular), position=pos)


def obj(ptype, material, translate, rot=None, scale=(1, 1, 1)):
    mv, inv = instance(translate, rot, scale)
    return R.make_object(ptype, material, mv, inv)


def custom_rays(starts, dirs):
    rays = np.zeros(len(starts), dtype=R.RAY_DTYPE)
    rays["start"][:, :3] = np.asarray(starts, dtype=F)
    rays["start"][:, 3] = 1.0
    rays["direction"][:, :3] = np.asarray(dirs, dtype=F)
    return rays


def main():
    if not oracle.reference_available():
        sys.exit("oracle/_ref is not built (needs /root/reference): run `make -C oracle` in the build container")

    # 1. the four sample scenes through the loader ------------------------------------------------------
    for scene, D in (("simpleSphere", 3), ("multipleSpheres", 3), ("simpleScene", 3), ("roundedCube", 5)):
        objs, lights = scene_loader.load_scene(str(SCENES / f"{scene}.txt"))
        for kernel in ("shade", "shade_and_reflect"):
            save(f"scene_{scene}_64_{kernel}", kernel, objs, lights, D, cam=(64, 64, 60.0))
        save(f"scene_{scene}_64_hittest", "hittest", objs, lights, 0, cam=(64, 64, 60.0))
    objs, lights = scene_loader.load_scene(str(SCENES / "simpleSphere.txt"))
    save("scene_simpleSphere_256_shade_and_reflect", "shade_and_reflect", objs, lights, 3, cam=(256, 256, 60.0))
    objs, lights = scene_loader.load_scene(str(SCENES / "roundedCube.txt"))
    save("scene_roundedCube_96x64_D3", "shade_and_reflect", objs, lights, 3, cam=(96, 64, 60.0))

    # 2. index recovery: light ambient 1, diffuse = specular = 0, material ambient encodes the index ------
    rng = np.random.default_rng(7)
    recs = []
    for i in range(40):
        amb = (((i + 1) & 255) / 256.0, (((i + 1) >> 8) & 255) / 256.0, ((i + 1) >> 16) / 256.0)
        t = R.SPHERE if i % 3 else R.BOX
        recs.append(obj(t, mat(ambient=amb), (rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-25, -8)),
                        rotation(rng.normal(size=3), rng.uniform(0, 6)), rng.uniform(0.5, 1.5, 3)))
    idx_objs = R.objects_array(recs)
    idx_light = R.lights_array([light((0, 0, 5, 1), ambient=(1, 1, 1), diffuse=(0, 0, 0), specular=(0, 0, 0))])
    save("index_recovery_shade", "shade", idx_objs, idx_light, 0, cam=(64, 64, 60.0))
    save("index_recovery_hittest", "hittest", idx_objs, idx_light, 0, cam=(64, 64, 60.0))

    # 3. hittest nearest-t: sphere-only / box-only -------------------------------------------------------
    o, l = random_scene(24, 0, 1, seed=11)
    save("hittest_spheres", "hittest", o, l, 0, cam=(64, 48, 60.0))
    o, l = random_scene(0, 24, 1, seed=12)
    save("hittest_boxes", "hittest", o, l, 0, cam=(64, 48, 60.0))

    # 4. tie-breaks: coincident spheres (later wins), coincident boxes (earlier wins), sphere == box hull ---
    one_light = R.lights_array([light((0, 0, 5, 1), ambient=(1, 1, 1), diffuse=(0, 0, 0), specular=(0, 0, 0))])
    m1, m2, m3 = mat(ambient=(1 / 256., 0, 0)), mat(ambient=(2 / 256., 0, 0)), mat(ambient=(3 / 256., 0, 0))
    save("tie_two_spheres", "shade", R.objects_array([obj(R.SPHERE, m1, (0, 0, -6)), obj(R.SPHERE, m2, (0, 0, -6))]),
         one_light, 0, cam=(32, 32, 60.0))
    save("tie_two_boxes", "shade", R.objects_array([obj(R.BOX, m1, (0, 0, -6), scale=(2, 2, 2)),
                                                     obj(R.BOX, m2, (0, 0, -6), scale=(2, 2, 2))]),
         one_light, 0, cam=(32, 32, 60.0))
    save("tie_sphere_box_sphere", "shade",
         R.objects_array([obj(R.SPHERE, m1, (0, 0, -6)), obj(R.BOX, m2, (0, 0, -6), scale=(2, 2, 2)),
                          obj(R.SPHERE, m3, (0, 0, -6))]), one_light, 0, cam=(32, 32, 60.0))

    # 5. multi-light: both kernels, both light orders (last light wins vs sum), stale specular ------------
    big = R.objects_array([obj(R.SPHERE, mat((1, 0, 0), (0, 1, 0), (0, 0, 1), shininess=8.0), (0, 0, -10), scale=(3, 3, 3))])
    l0 = light((10, 10, 0, 1))
    l1 = light((-8, 6, 2, 1), ambient=(.05, .05, .05), diffuse=(.2, .2, .2), specular=(.3, .3, .3))
    l2 = light((0, -12, -4, 1), ambient=(.1, .0, .1), diffuse=(.3, .1, .3), specular=(.1, .5, .1))
    for tag, ls in (("01", [l0, l1]), ("10", [l1, l0]), ("012", [l0, l1, l2]), ("210", [l2, l1, l0])):
        for kernel in ("shade", "shade_and_reflect"):
            save(f"lights_{tag}_{kernel}", kernel, big, R.lights_array(ls), 2, cam=(48, 48, 60.0))
    # stale specular (Q1b): shading normal != geometric normal (non-uniform scale; box edges) so that a light is
    # visible with nDotL <= 0 right after a light that produced specular
    stale = R.objects_array([
        obj(R.SPHERE, mat((.2, .1, .1), (.3, .6, .3), (1, 1, 1), shininess=4.0), (-2.5, 0, -9), rotation((0, 0, 1), .5), (2.5, .6, 1.2)),
        obj(R.BOX, mat((.1, .1, .2), (.3, .3, .6), (1, 1, 1), shininess=2.0), (2.5, 0, -9), rotation((1, 1, 0), .7), (2.5, 2.5, 2.5)),
    ])
    stale_lights = R.lights_array([light((6, 8, 2, 1)), light((-9, -3, -6, 1), ambient=(.1, .1, .1)),
                                   light((0, -10, -14, 1), ambient=(.05, .1, .05)), light((12, -2, -16, 1), ambient=(.1, .05, .05))])
    for kernel in ("shade", "shade_and_reflect"):
        save(f"stale_specular_{kernel}", kernel, stale, stale_lights, 2, cam=(96, 64, 60.0))

    # 6. bounce-loop edges: D x absorption, facing mirrors -------------------------------------------------
    for a in (1.0, 0.9995, 0.999, 0.5, 0.2):
        facing = R.objects_array([
            obj(R.SPHERE, mat((.3, .1, .1), (.5, .5, .2), (.8, .8, .8), absorption=a, shininess=10.0), (-2.2, 0, -10), scale=(2, 2, 2)),
            obj(R.SPHERE, mat((.1, .3, .1), (.2, .5, .5), (.8, .8, .8), absorption=a, shininess=10.0), (2.2, 0, -10), scale=(2, 2, 2)),
            obj(R.BOX, mat((.1, .1, .3), (.4, .4, .6), (.5, .5, .5), absorption=a, shininess=3.0), (0, -3.5, -10), scale=(12, 1, 8)),
        ])
        fl = R.lights_array([light((5, 12, 2, 1))])
        for D in (0, 1, 2, 3, 4):
            save(f"bounce_a{a}_D{D}", "shade_and_reflect", facing, fl, D, cam=(48, 32, 60.0))

    # 7. box edge cases: axis-parallel rays (dir == 0), origins on faces +-0.5, inside, corners ------------
    unit_box = R.objects_array([obj(R.BOX, mat((1 / 256., 0, 0), (.5, .5, .5), (.5, .5, .5)), (0, 0, 0))])  # mv = identity
    starts, dirs = [], []
    for sx in (-0.5, 0.5, -0.49999997, 0.49999997, 0.0, 0.25, -0.75, 0.75):
        for sy in (-0.5, 0.5, 0.0, 0.3):
            for d in ((0, 0, -1), (0, 0, 1), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (1, 1, 0), (0, 1, -1),
                      (1, 1, 1), (-1, -1, -1), (0.5, 0, -2)):
                for sz in (3.0, 0.0, 0.5, -0.5):
                    starts.append((sx, sy, sz))
                    dirs.append(d)
    edge_rays = custom_rays(starts, dirs)
    bl = R.lights_array([light((3, 4, 5, 1))])
    save("box_edges_hittest", "hittest", unit_box, bl, 0, rays=edge_rays)
    save("box_edges_shade", "shade", unit_box, bl, 0, rays=edge_rays)
    save("box_edges_shade_and_reflect", "shade_and_reflect", unit_box, bl, 2, rays=edge_rays)
    unit_sphere = R.objects_array([obj(R.SPHERE, mat((1 / 256., 0, 0), (.5, .5, .5), (.5, .5, .5)), (0, 0, 0))])
    save("sphere_edges_shade", "shade", unit_sphere, bl, 0, rays=edge_rays)   # origins inside / on the unit sphere (Q4)

    # 8. non-uniform scale + rotation (Q2), directional light (Q15), rays with w != canonical ---------------
    o, l = random_scene(6, 6, 3, seed=21, directional_lights=2)
    for kernel in ("shade", "shade_and_reflect"):
        save(f"directional_{kernel}", kernel, o, l, 3, cam=(64, 48, 60.0))
    odd = camera.primary_rays(48, 32, 60.0).copy()
    rng = np.random.default_rng(5)
    odd["start"][:, :3] = rng.uniform(-.5, .5, (len(odd), 3)).astype(F)
    odd["start"][:, 3] = rng.choice([1.0, 0.5, 2.0], len(odd)).astype(F)        # general start.w
    odd["direction"][:, 3] = rng.choice([0.0, 0.0, 0.01, -0.02], len(odd)).astype(F)  # general direction.w
    o, l = random_scene(8, 8, 2, seed=22)
    for kernel in ("hittest", "shade", "shade_and_reflect"):
        save(f"general_w_{kernel}", kernel, o, l, 2, rays=odd)

    # 9. random synthetic: 1k spheres, 4 lights, 32x32, D=3 (config-4 generator) ---------------------------
    o, l = synthetic.spheres_and_lights(1000, 4)
    for kernel in ("shade", "shade_and_reflect"):
        save(f"synthetic_1k_{kernel}", kernel, o, l, 3, cam=(32, 32, 60.0))
    o, l = random_scene(60, 40, 5, seed=31, directional_lights=1, spread=8.0)
    for kernel in ("shade", "shade_and_reflect"):
        save(f"random_mixed100_{kernel}", kernel, o, l, 3, cam=(64, 48, 60.0))

    # degenerate launches: no lights / no objects -----------------------------------------------------------
    o, l = random_scene(3, 3, 0, seed=41)
    save("no_lights_shade_and_reflect", "shade_and_reflect", o, l, 2, cam=(32, 32, 60.0))
    save("no_objects_shade", "shade", R.objects_array([]), R.lights_array([light((1, 1, 1, 1))]), 0, cam=(16, 16, 60.0))


def round2():
    """Fixtures added in round 2 (same generator, same reference kernels): the reference's SHIPPED bounce depth
    (MAX_BOUNCES = 30, OpenCL-Raytracer.cpp:75) on its own sample scenes and on the facing-mirror scene, NaN shadow
    rays, and degenerate instances whose NaN hit times make the loop's result depend on the object order."""
    # the reference's shipped workload is 2560x1440, fov 60, MAX_BOUNCES 30 (OpenCL-Raytracer.cpp:31-33,75): same
    # aspect ratio and depth at 1/16 of the resolution
    for scene in ("roundedCube", "simpleScene"):
        objs, lights = scene_loader.load_scene(str(SCENES / f"{scene}.txt"))
        save(f"shipped_{scene}_160x90_D30", "shade_and_reflect", objs, lights, 30, cam=(160, 90, 60.0))
    # unsigned post-decrement exhaustion (Q8, .cl:268,281) at odd / even / shipped depths
    for a in (0.5, 0.2):
        facing = R.objects_array([
            obj(R.SPHERE, mat((.3, .1, .1), (.5, .5, .2), (.8, .8, .8), absorption=a, shininess=10.0), (-2.2, 0, -10), scale=(2, 2, 2)),
            obj(R.SPHERE, mat((.1, .3, .1), (.2, .5, .5), (.8, .8, .8), absorption=a, shininess=10.0), (2.2, 0, -10), scale=(2, 2, 2)),
            obj(R.BOX, mat((.1, .1, .3), (.4, .4, .6), (.5, .5, .5), absorption=a, shininess=3.0), (0, -3.5, -10), scale=(12, 1, 8)),
        ])
        fl = R.lights_array([light((5, 12, 2, 1))])
        for D in (7, 30):
            save(f"bounce_a{a}_D{D}", "shade_and_reflect", facing, fl, D, cam=(48, 32, 60.0))
    # a mirror corridor: two big facing boxes with absorption .05 - paths really use all 30 bounces
    corridor = R.objects_array([
        obj(R.BOX, mat((.2, .2, .2), (.4, .4, .4), (.6, .6, .6), absorption=.05, shininess=20.0), (-3, 0, -12), scale=(1, 8, 10)),
        obj(R.BOX, mat((.2, .2, .2), (.4, .4, .4), (.6, .6, .6), absorption=.05, shininess=20.0), (3, 0, -12), scale=(1, 8, 10)),
        obj(R.SPHERE, mat((.6, .1, .1), (.6, .3, .3), (.9, .9, .9), absorption=.6, shininess=8.0), (0, 0, -12), scale=(.8, .8, .8)),
    ])
    save("corridor_D30", "shade_and_reflect", corridor, R.lights_array([light((0, 6, -6, 1)), light((0, -5, -2, 1), ambient=(.1, .1, .1))]),
         30, cam=(64, 40, 60.0))
    # NaN shadow rays: a directional light with a zero vector -> normalize(0) = NaN start, direction 0; every sphere /
    # box accepts the NaN time and `time >= 1 || time < 0` is false: blocked (shade_and_reflect_kernel.cl:197-209,229)
    o, _ = random_scene(5, 4, 0, seed=51)
    zero_dir = R.make_light(R.LightProperties((.2, .2, .2), (.6, .6, .6), (.8, .8, .8)), position=(0.0, 0.0, 0.0, 0.0))
    for tag, ls in (("last", [light((6, 8, 2, 1)), zero_dir]), ("first", [zero_dir, light((6, 8, 2, 1))])):
        for kernel in ("shade", "shade_and_reflect"):
            save(f"nan_shadow_{tag}_{kernel}", kernel, o, R.lights_array(ls), 2, cam=(48, 32, 60.0))
    # degenerate instances (what glm::inverse returns for a scale of 0: infinities / NaNs; and an all-zero 3x3 that
    # gives t = 0/0 for EVERY ray), last and in the middle of the object list: a NaN time overwrites and is overwritten
    base, l = random_scene(5, 3, 2, seed=52)
    def degenerate(kind):
        rec = obj(R.SPHERE, mat((.4, .4, .1), (.5, .5, .5), (.5, .5, .5), absorption=.6, shininess=6.0), (1.0, -0.5, -9.0))
        rec = rec.copy()
        inv = rec["mvInverse"].reshape(4, 4).copy()   # column-major: [c][r]
        if kind == "inf":
            inv[2, 2] = np.inf; inv[3, 2] = np.nan    # scale z = 0
            rec["mv"].reshape(4, 4)[2, 2] = 0.0
        else:
            inv[:3, :3] = 0.0                         # singular but finite
        rec["mvInverse"] = inv.reshape(16)
        return rec
    for kind in ("inf", "zero"):
        for where in ("last", "middle"):
            recs = list(base)
            recs.insert(len(recs) if where == "last" else 4, degenerate(kind))
            for kernel in ("shade", "shade_and_reflect"):
                save(f"degenerate_{kind}_{where}_{kernel}", kernel, R.objects_array(recs), l, 2, cam=(48, 32, 60.0))


if __name__ == "__main__":
    if not oracle.reference_available():
        sys.exit("oracle/_ref is not built (needs /root/reference): run `make -C oracle` in the build container")
    if "--round2" in sys.argv:
        round2()   # only the fixtures added in round 2 (the others are unchanged)
    else:
        main()
        round2()
