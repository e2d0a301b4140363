"""Drop-in for the two pure-tensor metrics of the reference's lib/metrics.py that the inversion and evaluation loops
call on device: ``psnr`` (30-45) and ``iou`` (79-94), same signatures, same asserts, one HIP launch each
(range check + clamp + reduction fused; no intermediate tensors).  ``ssim`` / LPIPS are CPU scikit-image / a VGG
network in the reference and are out of scope."""
import torch

from . import ops


def _range_ok(flag):
    # lib/metrics.py:22-27 range_check: the reference asserts on device tensors (a host synchronisation), so does this
    assert int(flag.item()) == 0, 'Range check failed'


def psnr(pred, target, reduction='mean'):
    assert pred.shape == target.shape
    assert len(pred.shape) == 4
    assert pred.shape[1] == 3 or pred.shape[-1] == 3  # Ensure RGB image
    batch_psnr, _, flag = ops.image_metrics(pred=pred.detach(), target=target.detach())
    _range_ok(flag)
    if reduction == 'mean':
        return batch_psnr.mean()
    return batch_psnr


def iou(alpha_pred, alpha_real, reduction='mean'):
    assert alpha_pred.shape == alpha_real.shape
    assert len(alpha_pred.shape) == 3 or (len(alpha_pred.shape) == 4 and alpha_pred.shape[1] == 1)
    _, batch_iou, flag = ops.image_metrics(mask_pred=alpha_pred.detach(), mask_real=alpha_real.detach())
    _range_ok(flag)
    if reduction == 'mean':
        return batch_iou.mean()
    return batch_iou.flatten()


def psnr_and_iou(pred, target, alpha_pred, alpha_real):
    """Both monitors of the inversion loop (run.py:2077-2090, 2247) in ONE launch: per-image (psnr [B], iou [B])."""
    batch_psnr, batch_iou, flag = ops.image_metrics(pred.detach(), target.detach(), alpha_pred.detach(), alpha_real.detach())
    _range_ok(flag)
    return batch_psnr, batch_iou
