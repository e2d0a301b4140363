"""Synthetic inversion (stand-in for BASELINE config 3): the HIP renderer and the oracle renderer, each
driving Adam on latent + pose with the same noise, must follow the same PSNR / IoU trajectory."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
pytestmark = pytest.mark.gpu


def test_inversion_trajectories_match(gpu_device):
    import inversion_synthetic as inv
    h_hip, h_ref, _, _ = inv.run(gpu_device, res=32, samples=32, batch=2, steps=12, plane_res=48)
    for (p_h, i_h, l_h), (p_r, i_r, l_r) in zip(h_hip, h_ref):
        # dB.  Both runs are fp32 with different summation orders (and Adam divides by the gradient's running
        # magnitude), so the two trajectories drift apart slowly; measured on MI355X: <= 0.16 dB over these 12 steps
        assert abs(p_h - p_r) < 0.3, (p_h, p_r)
        assert abs(i_h - i_r) < 0.02, (i_h, i_r)
    assert h_hip[-1][2] < h_hip[0][2], 'the loss did not go down'
    assert h_hip[-1][0] > h_hip[0][0], 'PSNR did not improve'
