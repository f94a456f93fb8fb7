"""MI355X-native backend for the reference's IRaytracer hot path.

Host-side mirror of the reference interface (Python flavour; the C++ flavour is host/HIPRaytracer.hpp):
  records       device-layout ObjectData / Light / Ray records + float32 matrix toolkit
  scene_loader  SceneLoader::Load
  camera        primary-ray generation (main()'s ray loop)
  ppm           PPMExporter::ExportP3
  synthetic     the synthetic BASELINE configuration (100k spheres / 32 lights)
  hip_raytracer HIPRaytracer(objects, lights, rays, MAX_BOUNCES).Render() over the C-ABI library

Nothing here falls back to a CPU implementation: constructing a HIPRaytracer without the built
HIP library (csrc/libhip_raytracer.so) or without a GPU raises.
"""
from . import camera, ppm, records, scene_loader, synthetic  # noqa: F401

__all__ = ["camera", "ppm", "records", "scene_loader", "synthetic"]
