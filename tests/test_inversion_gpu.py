"""Synthetic inversion (stand-in for BASELINE config 3): the HIP renderer and the oracle renderer, each
driving Adam on latent + pose with the same noise, must follow the same PSNR / IoU trajectory, and the HIP gradients
must match the oracle's at every point of the ORACLE's trajectory."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
pytestmark = pytest.mark.gpu


def test_inversion_trajectories_match(gpu_device):
    import inversion_synthetic as inv
    h_hip, h_ref, _, _, along, h_pert, cond = inv.run(gpu_device, res=32, samples=32, batch=2, steps=12, plane_res=48,
                                                     teacher=True)
    # (1) no chaos involved: HIP loss / gradients evaluated at the oracle's own parameters, every step.  The loss is a
    # mean-squared error over 6144 values, i.e. upstream gradients of 1e-6: the case that exposed the unscaled
    # split-fp16 gradient operands in round 2 (3-8 % error here while every unit-scale gradient test passed).
    for l_hip, l_ref, e_ws, e_delta in along:
        assert abs(l_hip - l_ref) <= 2e-6 * abs(l_ref) + 1e-9, (l_hip, l_ref)
        assert e_ws < 1e-4 and e_delta < 2e-3, (e_ws, e_delta)
    # ... and at the start point against the float64 oracle, where the float32 oracle itself is off by cond['f32']
    assert cond['hip'][0] < max(1e-4, 4 * cond['f32'][0]) and cond['hip'][1] < max(1e-3, 4 * cond['f32'][1]), cond
    # (2) free-running trajectories.  Adam normalises the update, so the loop is chaotic: the oracle itself, with a
    # relative perturbation of 1e-4 on its gradients, is 0.6 dB away after these 12 steps (h_pert).  Measured on
    # MI355X after the fix: 1e-4 dB between HIP and oracle; the bound is the perturbed oracle's drift.
    drift = max(abs(a[0] - b[0]) for a, b in zip(h_pert, h_ref))
    for (p_h, i_h, l_h), (p_r, i_r, l_r) in zip(h_hip, h_ref):
        assert abs(p_h - p_r) < max(0.05, drift), (p_h, p_r, drift)
        assert abs(i_h - i_r) < 0.02, (i_h, i_r)
    assert h_hip[-1][2] < h_hip[0][2], 'the loss did not go down'
    assert h_hip[-1][0] > h_hip[0][0], 'PSNR did not improve'
