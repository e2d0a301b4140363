"""Drop-in for the reference's ``lib/pose_estimation.py`` (``compute_pose_pnp`` 30-131, ``get_focal_guesses`` 134-143)
and for ``estimate_poses_batch`` of run.py:1709-1740 - on the GPU, without the ``.cpu().numpy()`` round trip and the
per-image, per-focal-proposal OpenCV calls that serialise every inversion batch of the reference.

    import nerf_from_image_amd.pose_estimation as pose_estimation        # replaces `from lib import pose_estimation`
    cam2world, focal, errors = pose_estimation.estimate_poses_batch(coords, mask, focal_guesses)

Same arguments, same return meaning; tensors stay on the device (the reference returns numpy arrays from
compute_pose_pnp and moves them back).  The solver is NOT OpenCV's SQPNP: OpenCV is absent from this image, so what is
restated is the problem (reprojection-error minimisation per focal proposal, smallest RMS error with t_z > 0 wins), solved
by a normalised DLT start + Levenberg-Marquardt in float64 inside one kernel launch per batch (csrc/nfi_pnp.inc).
PARITY UNPINNED: no OpenCV outputs exist to pin against; tests check pose recovery on synthetic correspondences and
agreement with an independent float64 solver kept with the test infrastructure.
"""
import torch

from . import _lib, ops
from .augment import invert_space

REFINE_ITERATIONS = 30


def get_focal_guesses(focal_length):
    """Focal-length proposals for the sweep: the 1st, 10th ... 90th, 99th percentiles of the dataset's focal lengths,
    duplicates removed (pose_estimation.py:134-143); None for orthographic datasets."""
    if focal_length is None:
        return None
    f = focal_length.detach().double().flatten()
    q = torch.tensor([1, 10, 20, 30, 40, 50, 60, 70, 80, 90, 99], dtype=torch.float64, device=f.device) / 100.0
    return torch.unique(torch.quantile(f, q, interpolation='linear'))


def compute_pose_pnp(coords, masks, focal_proposals, refine=True):
    """coords [B,H,W,3] canonical coordinates, masks [B,H,W] (bool, or float: foreground where > 0.5), focal_proposals:
    sequence / array / tensor of focal lengths.  Returns (world2cam [B,4,4], focal [B], errors [B]) float32 on the
    device: ``flip @ [R | t]`` per image for the best proposal, the reference's dummy pose where nothing solves."""
    if not coords.is_cuda:
        raise RuntimeError('compute_pose_pnp: coords must live on the GPU (no CPU path in nerf_from_image_amd)')
    coords = ops._f32c(coords.float(), 'coords')
    B, H, W, three = coords.shape
    if three != 3 or tuple(masks.shape) != (B, H, W):
        raise ValueError('compute_pose_pnp: coords [B,H,W,3] and masks [B,H,W] expected, got %s and %s' % (
            tuple(coords.shape), tuple(masks.shape)))
    mask = masks.to(device=coords.device, dtype=torch.float32).contiguous()
    focals = torch.as_tensor(focal_proposals, dtype=torch.float32).flatten().to(coords.device).contiguous()
    nf = focals.numel()
    if nf == 0:
        raise ValueError('compute_pose_pnp: no focal proposals')
    lib = _lib.load()
    ws = torch.empty((lib.nfi_pnp_workspace_bytes(B, nf) // 8,), dtype=torch.float64, device=coords.device)
    world2cam = torch.empty((B, 4, 4), dtype=torch.float32, device=coords.device)
    focal = torch.empty((B,), dtype=torch.float32, device=coords.device)
    error = torch.empty((B,), dtype=torch.float32, device=coords.device)
    with torch.cuda.device(coords.device):
        _lib.call_struct('nfi_pose_pnp', 'nfi_pnp_args', ops._stream(coords), n_images=B, height=H, width=W, coords=coords,
                         mask=mask, mask_threshold=0.5, n_focal=nf, focal_proposals=focals,
                         refine_iterations=REFINE_ITERATIONS if refine else 0, workspace=ws, workspace_bytes=ws.numel() * 8,
                         world2cam=world2cam, focal=focal, error=error)
    return world2cam, focal, error


def estimate_poses_batch(target_coords, target_mask, focal_guesses):
    """run.py:1709-1740.  target_coords [B,H,W,3], target_mask [B,H,W] (soft mask, foreground above 0.9),
    focal_guesses: proposals or None (orthographic dataset: solved with a very long focal length and converted back).
    Returns (cam2world [B,4,4], focal [B] or None, errors [B])."""
    foreground = target_mask > 0.9
    orthographic = focal_guesses is None
    long_focal = 100.0
    world2cam, focal, errors = compute_pose_pnp(target_coords, foreground, [long_focal] if orthographic else focal_guesses)
    if not orthographic:
        return invert_space(world2cam), focal, errors
    # perspective with f = 100 approximates the orthographic camera: depth -> scale, image-plane shift rescaled, the
    # camera pushed back to z = -10 (run.py:1725-1737)
    scale = 2 * long_focal / -world2cam[:, 2, 3]
    w2c = world2cam.clone()
    w2c[:, :2, 3] = world2cam[:, :2, 3] * scale[:, None]
    w2c[:, 2, 3] = -10.0
    return invert_space(w2c) / scale[:, None, None], None, errors
