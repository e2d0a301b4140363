"""Pose from a canonical-coordinate map (lib/pose_estimation.py:30-131 + run.py:1709-1740).

PARITY UNPINNED (cv2 is not available: no reference outputs exist).  CPU: the independent float64 solver of
oracle/nfi_oracle_pnp.py recovers known poses from exact synthetic correspondences and picks the right focal proposal.
GPU: the HIP kernel (through the C ABI and the drop-in module) recovers the same poses to < 0.5 degrees / 1e-3, agrees
with the oracle on noisy input, and reproduces the reference's conventions (flip, dummy pose, orthographic conversion)."""
import numpy as np
import pytest
import torch

from oracle import nfi_oracle_pnp as orp


def rot_angle_deg(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1) / 2
    return float(np.degrees(np.arccos(np.clip(c, -1, 1))))


FLIP = np.diag([1.0, -1.0, -1.0, 1.0])


def test_oracle_recovers_known_poses_and_focal():
    coords, masks, Rs, ts = orp.synthetic_correspondences(3, 48, seed=0, focal=1.2)
    w2c, focal, err = orp.compute_pose_pnp(coords, masks, [0.8, 1.0, 1.2, 1.5])
    assert np.allclose(focal, 1.2) and err.max() < 1e-5
    for b in range(3):
        m = FLIP @ w2c[b]
        assert rot_angle_deg(m[:3, :3], Rs[b]) < 0.05 and np.abs(m[:3, 3] - ts[b]).max() < 1e-3
    # no foreground pixels (fewer than 4: pose_estimation.py:58): the reference's dummy pose
    masks[0] = False
    w2c, focal, err = orp.compute_pose_pnp(coords[:1], masks[:1], [1.2])
    assert np.allclose(w2c[0], FLIP @ np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, -10.0], [0, 0, 0, 1]])) and err[0] == 10.0


@pytest.mark.gpu
def test_hip_pose_recovery(gpu_device):
    import nerf_from_image_amd.pose_estimation as pe
    coords, masks, Rs, ts = orp.synthetic_correspondences(4, 64, seed=1, focal=1.1)
    proposals = [0.7, 0.9, 1.1, 1.3, 1.6]
    w2c, focal, err = pe.compute_pose_pnp(torch.from_numpy(coords).to(gpu_device), torch.from_numpy(masks).to(gpu_device), proposals)
    w2c, focal, err = w2c.cpu().double().numpy(), focal.cpu().numpy(), err.cpu().numpy()
    assert np.allclose(focal, 1.1) and err.max() < 1e-4, (focal, err)
    for b in range(4):
        m = FLIP @ w2c[b]
        assert rot_angle_deg(m[:3, :3], Rs[b]) < 0.5 and np.abs(m[:3, 3] - ts[b]).max() < 1e-3, (b, m, Rs[b], ts[b])
    # without refinement the linear start alone is already close on exact data
    w2c0, _, err0 = pe.compute_pose_pnp(torch.from_numpy(coords).to(gpu_device), torch.from_numpy(masks).to(gpu_device), [1.1],
                                        refine=False)
    assert float(err0.max()) < 1e-3


@pytest.mark.gpu
def test_hip_pose_matches_independent_solver_on_noisy_input(gpu_device):
    import nerf_from_image_amd.pose_estimation as pe
    coords, masks, Rs, ts = orp.synthetic_correspondences(3, 64, seed=2, noise=0.02, focal=1.0)
    masks[2] = False
    masks[2, 0, :3] = True                                    # 3 foreground pixels: nothing to solve
    proposals = np.array([0.8, 1.0, 1.25])
    ref_w2c, ref_f, ref_e = orp.compute_pose_pnp(coords, masks, proposals)
    w2c, focal, err = pe.compute_pose_pnp(torch.from_numpy(coords).to(gpu_device), torch.from_numpy(masks).to(gpu_device), proposals)
    assert np.allclose(focal.cpu().numpy(), ref_f)
    assert np.allclose(err.cpu().numpy(), ref_e, rtol=1e-4, atol=1e-6), (err, ref_e)
    assert np.abs(w2c.cpu().double().numpy() - ref_w2c).max() < 2e-4
    assert float(err[2]) == 10.0 and float(focal[2]) == 1.0


def few_point_case(n, seed, planar=False, res=16, focal=1.2):
    """n pixels of a res x res image whose canonical coordinates are exact for a known pose; planar: the points lie in a
    plane (the linear start of the solvers is degenerate there)."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    R = Rotation.from_rotvec(rng.normal(size=3) * 0.9).as_matrix()
    t = np.array([rng.normal() * 0.05, rng.normal() * 0.05, 2.5])
    pix = rng.choice(res * res, size=n, replace=False)
    xy = orp.screen_grid(res, res)[pix]
    if planar:
        nrm, c = np.array([0.3, -0.2, 1.0]), 2.5
        z = c / (nrm[0] * xy[:, 0] / focal + nrm[1] * xy[:, 1] / focal + nrm[2])
    else:
        z = rng.uniform(2.0, 3.0, size=n)
    P_cam = np.stack([xy[:, 0] * z / focal, xy[:, 1] * z / focal, z], axis=1)
    coords = np.zeros((1, res * res, 3), np.float32)
    coords[0, pix] = (P_cam - t) @ R                          # R^T (P_cam - t)
    masks = np.zeros((1, res * res), bool)
    masks[0, pix] = True
    return coords.reshape(1, res, res, 3), masks.reshape(1, res, res), R, t


def test_oracle_solves_few_and_coplanar_points():
    """4 or 5 pixels (the reference solves from 4 on) and coplanar points (no linear start): the multi-start refinement."""
    for n, planar, seed in ((5, False, 0), (12, True, 2)):
        coords, masks, R, t = few_point_case(n, seed, planar)
        w2c, focal, err = orp.compute_pose_pnp(coords, masks, [1.2])
        m = FLIP @ w2c[0]
        assert err[0] < 1e-5 and rot_angle_deg(m[:3, :3], R) < 0.1 and np.abs(m[:3, 3] - t).max() < 2e-3, (n, planar, err, m, R, t)
    coords, masks, R, t = few_point_case(4, 5)
    w2c, focal, err = orp.compute_pose_pnp(coords, masks, [1.2])
    assert err[0] < 1e-5 and (FLIP @ w2c[0])[2, 3] > 0           # four points: A solution (there can be several)


@pytest.mark.gpu
def test_hip_solves_few_and_coplanar_points(gpu_device):
    import nerf_from_image_amd.pose_estimation as pe
    for n, planar, seed in ((5, False, 0), (5, False, 1), (12, True, 2), (40, True, 3), (6, False, 4)):
        coords, masks, R, t = few_point_case(n, seed, planar)
        w2c, focal, err = pe.compute_pose_pnp(torch.from_numpy(coords).to(gpu_device), torch.from_numpy(masks).to(gpu_device), [1.0, 1.2])
        m = FLIP @ w2c[0].cpu().double().numpy()
        assert float(err[0]) < 1e-4 and float(focal[0]) == pytest.approx(1.2), (n, planar, err, focal)
        assert rot_angle_deg(m[:3, :3], R) < 0.5 and np.abs(m[:3, 3] - t).max() < 5e-3, (n, planar, m, R, t)
    coords, masks, R, t = few_point_case(4, 5)
    w2c, focal, err = pe.compute_pose_pnp(torch.from_numpy(coords).to(gpu_device), torch.from_numpy(masks).to(gpu_device), [1.2])
    assert float(err[0]) < 1e-4 and float((FLIP @ w2c[0].cpu().double().numpy())[2, 3]) > 0
    coords, masks, R, t = few_point_case(3, 6)
    w2c, focal, err = pe.compute_pose_pnp(torch.from_numpy(coords).to(gpu_device), torch.from_numpy(masks).to(gpu_device), [1.2])
    assert float(err[0]) == 10.0                               # three points: the dummy pose, like the reference


@pytest.mark.gpu
def test_estimate_poses_batch_conventions(gpu_device):
    """run.py:1709-1740: soft mask threshold 0.9, cam2world = invert_space(flip @ [R|t]); the orthographic branch solves
    with f = 100 and converts depth into the orthographic scale."""
    import nerf_from_image_amd.pose_estimation as pe
    coords, masks, Rs, ts = orp.synthetic_correspondences(2, 64, seed=3, focal=1.3)
    soft = torch.from_numpy(masks).float().to(gpu_device) * 0.95 + 0.02
    c2w, focal, err = pe.estimate_poses_batch(torch.from_numpy(coords).to(gpu_device), soft, pe.get_focal_guesses(
        torch.tensor([1.3] * 7 + [0.9, 1.8])))
    assert focal is not None and torch.allclose(focal.cpu(), torch.tensor([1.3, 1.3]))
    for b in range(2):
        w2c = np.eye(4)
        w2c[:3, :3], w2c[:3, 3] = Rs[b], ts[b]
        expect = np.linalg.inv(FLIP @ w2c)
        assert np.abs(c2w[b].cpu().double().numpy() - expect).max() < 2e-3
    # orthographic: a distant camera, f = 100
    coords_o, masks_o, Ro, to = orp.synthetic_correspondences(2, 64, seed=4, focal=100.0, distance=220.0)
    c2w_o, focal_o, _ = pe.estimate_poses_batch(torch.from_numpy(coords_o).to(gpu_device), torch.from_numpy(masks_o).float().to(gpu_device), None)
    assert focal_o is None and c2w_o.shape == (2, 4, 4) and bool(torch.isfinite(c2w_o).all())
    ref_w2c, _, _ = orp.compute_pose_pnp(coords_o, masks_o, [100.0])
    for b in range(2):                                         # the same conversion on the independent solver's pose
        s_ = 2 * 100.0 / -ref_w2c[b, 2, 3]
        w = ref_w2c[b].copy()
        w[:2, 3] *= s_
        w[2, 3] = -10.0
        inv = np.eye(4)
        inv[:3, :3] = w[:3, :3].T / w[3, 3]
        inv[:3, 3] = -(w[:3, :3].T / w[3, 3]) @ w[:3, 3]
        expect = inv / s_
        assert np.abs(c2w_o[b].cpu().double().numpy() - expect).max() < 2e-3 * np.abs(expect).max(), (c2w_o[b], expect)
        assert rot_angle_deg((FLIP @ ref_w2c[b])[:3, :3], Ro[b]) < 0.5
    g = pe.get_focal_guesses(torch.tensor([1.0, 2.0, 3.0, 4.0]))
    assert torch.allclose(g, torch.from_numpy(np.unique(np.percentile(np.array([1.0, 2, 3, 4]), [1, 10, 20, 30, 40, 50, 60, 70, 80, 90, 99]))))
    assert pe.get_focal_guesses(None) is None
