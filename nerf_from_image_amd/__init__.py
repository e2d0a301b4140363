"""MI355X-native volumetric-rendering hot path of nerf-from-image (HIP kernels behind a C ABI).

Public surface (mirrors the reference's own callables for this path):
  nerf_from_image_amd.nerf_utils   <- lib/nerf_utils.py
  nerf_from_image_amd.generator    <- the sampler closure / TriplanarDecoder part of models/generator.py
  nerf_from_image_amd.render       <- run.py::render
  nerf_from_image_amd.ops          <- tensor-level wrappers over include/nfi_hip.h
"""
__all__ = ['ops']
