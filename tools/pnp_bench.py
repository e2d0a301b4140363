"""Timing of the device pose solver (nfi_pose_pnp) on an inversion-sized batch: 16 images x 128x128 canonical-coordinate maps,
11 focal proposals (lib/pose_estimation.py:134-143), 30 Levenberg-Marquardt passes - against the independent float64
CPU solver on the same input (the reference's own path is OpenCV on the host, absent here)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import nfi_oracle_pnp as orp  # noqa: E402
import nerf_from_image_amd.pose_estimation as pe  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    coords, masks, Rs, ts = orp.synthetic_correspondences(16, 128, seed=5, noise=0.01, focal=1.1)
    proposals = np.linspace(0.8, 1.4, 11)
    c, m = torch.from_numpy(coords).to(dev), torch.from_numpy(masks).to(dev)
    pe.compute_pose_pnp(c, m, proposals)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        w2c, focal, err = pe.compute_pose_pnp(c, m, proposals)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / 10 * 1e3
    t0 = time.perf_counter()
    ref = orp.compute_pose_pnp(coords[:2], masks[:2], proposals)
    cpu_ms = (time.perf_counter() - t0) * 1e3 / 2 * 16
    print('pose_pnp: 16 images x 128x128 px x 11 focal proposals: HIP %.2f ms per batch (2 launches); independent float64 CPU solver '
          '%.0f ms per batch (extrapolated from 2 images, 1 thread); chosen focal agrees on the 2 images: %s; max |error diff| %.2e'
          % (gpu_ms, cpu_ms, bool(np.allclose(focal[:2].cpu().numpy(), ref[1])), float(np.abs(err[:2].cpu().numpy() - ref[2]).max())))


if __name__ == '__main__':
    main()
