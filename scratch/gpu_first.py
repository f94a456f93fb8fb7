"""First GPU shake-out: HIP vs restatement on several scenes (scratch, not a test)."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from helpers import *
from oracle import oracle
from opencl_raytracer_amd import scene_loader, synthetic
from opencl_raytracer_amd.hip_raytracer import HIPRaytracer

def check(name, objs, lights, rays, D, kernels=('hittest','shade','shade_and_reflect')):
    for fused in (True, False):
        rs = oracle.Restatement(fused)
        for k in kernels:
            o = rs.render(k, objs, lights, rays, D)
            for literal in (False, True):
                for raygen in (True, False):
                    rt = HIPRaytracer(objs, lights, rays, D, kernel=k, fused=fused, literal=literal, raygen=raygen)
                    out = rt.Render()
                    t, idx = rt.render_aux()
                    st = rt.count_rays()
                    if k == 'hittest':
                        bit = np.array_equal(out.view(np.uint32), o['out'].view(np.uint32)); md = 0.0
                    else:
                        bit = np.array_equal(rgb_bits(out), rgb_bits(o['out'])); md = compare_frames(out, o['out'])
                    print(f"{name:14s} {k:18s} fused={int(fused)} lit={int(literal)} gen={int(raygen)} pin={st.pinhole} "
                          f"bitexact={bit} maxd={md:.2e} idx_eq={np.array_equal(idx, o['hit_index'])} "
                          f"t_eq={np.array_equal(t.view(np.uint32), o['hit_t'].view(np.uint32))} "
                          f"Rref={st.rays_reference}/{o['rays_ref']} Ract={st.rays_traced} hits={st.hit_pixels}")
                    rt.close()

for sc, (w, h) in (('simpleSphere', (64, 64)), ('multipleSpheres', (96, 64)), ('simpleScene', (64, 64)), ('roundedCube', (64, 64))):
    objs, lights = scene_loader.load_scene(f'/root/repo/scenes/{sc}.txt')
    check(sc, objs, lights, camera.primary_rays(w, h), 3)
objs, lights = random_scene(20, 12, 4, seed=1, directional_lights=1)
check('random32', objs, lights, camera.primary_rays(64, 48), 3)
objs, lights = synthetic.spheres_and_lights(1000, 4)
check('synth1k', objs, lights, camera.primary_rays(32, 32), 3, kernels=('shade_and_reflect',))
