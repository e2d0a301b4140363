"""Drop-in for ``augment`` / ``augment_impl`` of the reference's run.py (720-815): same arguments, same returns, the
same random draws in the same order (torch.rand / torch.randn of [bs] tensors on the image's device).

The image part - F.affine_grid + F.grid_sample over [bs,C,H,W], 15 augmented copies of cat(prediction, target) per
inversion step (run.py:2216-2231) - is ONE HIP launch forward and one backward (nfi_affine_warp_fwd/bwd build the
2x3 matrix from the draws themselves).  The pose part is 4x4 matrix algebra on [bs] poses and stays tensor
expressions.

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
    """cam2world <-> world2cam for [B,4,4] similarity matrices [[R, t], [0, s]] (what lib/pose_utils.py:20-27 computes):
    the inverse is [[(R/s)^T, -(R/s)^T t], [0, 1]].  Same elementwise operations as the reference (division by s,
    product with t, sum over the row index), so the result is bit-identical."""
    s = mat[:, 3:4, 3:4]
    rot_over_s = mat[:, :3, :3] / s
    shift = -(rot_over_s * mat[:, :3, 3].unsqueeze(-1)).sum(dim=1)
    top = torch.cat([rot_over_s.transpose(1, 2), shift.unsqueeze(-1)], dim=2)
    bottom = torch.zeros_like(mat[:, 3:4, :])
    bottom[:, 0, 3] = 1
    return torch.cat([top, bottom], dim=1)


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


def draw_transform(bs, device, p, disable_scale):
    """The random in-plane rotation / zoom / shift of one augmentation (run.py:724-741), each applied with probability
    p.  The DRAW ORDER is the reference's - rand (angle), rand (coin), [randn (zoom), rand (coin)], randn (shift),
    rand (coin) - so a seeded run consumes the generator exactly like run.py; a coin that comes up tails selects the
    identity value (torch.where gives the same bits as the reference's lerp with weights 0 / 1)."""
    def coin(*shape):
        return torch.rand(shape, device=device) < p

    angle = (torch.rand((bs,), device=device) - 0.5) * 2 * np.pi
    angle = angle * coin(bs).float()
    zoom = torch.ones((bs,), device=device)
    if not disable_scale:
        drawn = torch.exp2(torch.randn((bs,), device=device) * 0.2)
        zoom = torch.where(coin(bs), drawn, zoom)
    drawn = torch.randn((bs, 2), device=device) * 0.1
    shift = torch.where(coin(bs, 1), drawn, torch.zeros_like(drawn))
    return angle, zoom, shift


def augment_impl(img, pose, focal, p, disable_scale=False, cached_tform=None):
    if dataset_config is None:
        raise RuntimeError('nerf_from_image_amd.augment.configure(args, dataset_config) has not been called')
    bs = img.shape[0] if img is not None else pose.shape[0]
    device = img.device if img is not None else pose.device

    rot, scale, translation = draw_transform(bs, device, p, disable_scale) if cached_tform is None else cached_tform
    cached_tform = rot, scale, translation

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
    """run.py:798-815: identity when nothing is to be drawn or replayed; otherwise augment_impl, with the drawn
    transform appended on request."""
    if cached_tform is None and p == 0:
        return img, pose, focal
    if img is not None and pose is not None:
        assert img.shape[0] == pose.shape[0]
    result = augment_impl(img, pose, focal, p, disable_scale, cached_tform)
    return result if return_tform else result[:3]
