"""Drop-in for ``augment`` / ``augment_impl`` of the reference's run.py (720-815): same arguments, same returns, the
same random draws in the same order (torch.rand / torch.randn of [bs] tensors on the image's device).

The image part - F.affine_grid + F.grid_sample over [bs,C,H,W], 15 augmented copies of cat(prediction, target) per
inversion step (run.py:2216-2231) - is ONE HIP launch forward and one backward (nfi_affine_warp_fwd/bwd build the
2x3 matrix from the draws themselves).  The pose part is 4x4 matrix algebra on [bs] poses and stays tensor
expressions (lib/pose_utils.invert_space restated below).

    import nerf_from_image_amd.augment as nfi_aug
    nfi_aug.configure(args, dataset_config)        # reads dataset_config['white_background'], args.supervise_alpha
    augment = nfi_aug.augment                       # replaces run.py's own def
"""
import numpy as np
import torch

from . import ops
from .autograd import differentiable

args = None
dataset_config = None


def configure(new_args, new_dataset_config):
    global args, dataset_config
    args, dataset_config = new_args, new_dataset_config


def invert_space(mat):
    """lib/pose_utils.py:20-27: cam2world <-> world2cam for [B,4,4] matrices with a scale in [3,3]."""
    out_mat = torch.zeros_like(mat)
    out_mat[:, :3, :3] = mat[:, :3, :3].transpose(-2, -1) / mat[:, 3:4, 3:4]
    out_mat[:, 3, 3] = 1
    out_mat[:, :3, 3] = -torch.sum(mat[:, :3, :3] / mat[:, 3:4, 3:4] * mat[:, :3, None, 3], dim=-2)
    return out_mat


def warp_images(img, rot, scale, translation, white_background):
    """The image branch of augment_impl (run.py:745-766) as one differentiable HIP op (gradient w.r.t. img only: the
    transform parameters are random draws)."""
    r, s, t = rot.detach(), None if scale is None else scale.detach(), translation.detach()

    def fwd(x):
        return ops.affine_warp(x, r, s, t, white_background)

    def bwd(inputs, out_meta, grads, needs):
        return (ops.affine_warp_bwd(grads[0].contiguous(), r, s, t, white_background),)
    return differentiable('affine_warp', fwd, img, bwd=bwd)


def augment_pose(pose, focal, rot, scale, translation):
    """Camera-side counterpart of the image warp (run.py:771-791), for [bs,4,4] cam2world poses.

    The in-plane rotation turns the camera about its viewing axis; the zoom goes into the focal length (perspective)
    or into the orthographic extent - the rotation block and pose[3,3] (focal None); the 2-D shift moves the camera in
    its own image plane, by translation * depth / (2 focal) (perspective) or translation * extent (orthographic),
    which is easiest to state on the world2cam matrix."""
    bs = pose.shape[0]
    c, s = torch.cos(rot), torch.sin(rot)
    spin_t = torch.eye(4, device=pose.device).repeat(bs, 1, 1)       # transpose of [[c, -s], [s, c]] in the x/y block
    spin_t[:, 0, 0], spin_t[:, 0, 1], spin_t[:, 1, 0], spin_t[:, 1, 1] = c, s, -s, c
    pose = pose @ spin_t
    orthographic = focal is None
    if orthographic:
        pose[:, :3, :3] *= scale[:, None, None]
        pose[:, 3:4, 3:4] *= scale[:, None, None]
    else:
        focal = focal / scale
    extent = pose[:, 3:4, 3]                                         # [bs,1] orthographic scale entry
    world2cam = invert_space(pose)
    if orthographic:
        world2cam[:, :2, 3] -= translation * extent
    else:
        world2cam[:, :2, 3] -= translation * (-world2cam[:, 2:3, 3] / (2 * focal[:, None]))
    out = invert_space(world2cam)
    if orthographic:
        out[:, :3, :3] *= pose[:, 3:4, 3:4]
        out[:, 3, 3] *= pose[:, 3, 3]
    return out, focal


def augment_impl(img, pose, focal, p, disable_scale=False, cached_tform=None):
    if dataset_config is None:
        raise RuntimeError('nerf_from_image_amd.augment.configure(args, dataset_config) has not been called')
    bs = img.shape[0] if img is not None else pose.shape[0]
    device = img.device if img is not None else pose.device

    if cached_tform is None:
        rot = (torch.rand((bs,), device=device) - 0.5) * 2 * np.pi
        rot = rot * (torch.rand((bs,), device=device) < p).float()
        if disable_scale:
            scale = torch.ones((bs,), device=device)
        else:
            scale = torch.exp2(torch.randn((bs,), device=device) * 0.2)
            scale = torch.lerp(torch.ones_like(scale), scale, (torch.rand((bs,), device=device) < p).float())
        translation = torch.randn((bs, 2), device=device) * 0.1
        translation = torch.lerp(torch.zeros_like(translation), translation,
                                 (torch.rand((bs, 1), device=device) < p).float())
        cached_tform = rot, scale, translation
    else:
        rot, scale, translation = cached_tform

    if img is not None:
        white = bool(dataset_config['white_background'])
        if white:
            assert not args.supervise_alpha
        img_transformed = warp_images(img, rot, scale, translation, white)
    else:
        img_transformed = None

    if pose is not None:
        pose, focal = augment_pose(pose, focal, rot, scale, translation)

    return img_transformed, pose, focal, cached_tform


def augment(img, pose, focal, p, disable_scale=False, cached_tform=None, return_tform=False):
    if p == 0 and cached_tform is None:
        return img, pose, focal
    assert img is None or pose is None or img.shape[0] == pose.shape[0]
    img_new, pose_new, focal_new, tform = augment_impl(img, pose, focal, p, disable_scale, cached_tform)
    if return_tform:
        return img_new, pose_new, focal_new, tform
    return img_new, pose_new, focal_new
