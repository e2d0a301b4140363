"""TEST INFRASTRUCTURE ONLY - CPU restatement of the renderer's neighbours in the inversion loop.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product package
(nerf_from_image_amd/) never does.

  warp_images   the image branch of run.py::augment_impl (run.py:745-766): 2x3 matrix from (rot, scale, translation),
                F.affine_grid(align_corners=False) + F.grid_sample(bilinear, zeros, align_corners=False), with the
                white-background adjustment;
  psnr / iou    lib/metrics.py:30-45 / 79-94, per image;
  torgb_upsample_add   the tail of the last synthesis block (models/stylegan.py:424-433): upsample2d (72-76) of the
                previous image + OutputLayer (351-372: 1x1 modulated conv without demodulation + bias).

Pinned: oracle/make_golden.py (neighbours()) drives the LIVE functions (AST-sliced out of run.py / lib/metrics.py,
which are not importable as modules) and asserts bit-equality with these restatements before it writes
tests/golden/neighbours.npz.  The numerics live in ATen (F.affine_grid, F.grid_sample, reductions)."""
import torch
import torch.nn.functional as F


def warp_matrix(rot, scale, translation):
    """mat_scaled of run.py:738-752: [bs,2,3]."""
    bs = rot.shape[0]
    mat = torch.zeros((bs, 2, 3), dtype=rot.dtype, device=rot.device)
    mat[:, 0, 0] = torch.cos(rot)
    mat[:, 0, 1] = -torch.sin(rot)
    mat[:, 0, 2] = translation[:, 0]
    mat[:, 1, 0] = torch.sin(rot)
    mat[:, 1, 1] = torch.cos(rot)
    mat[:, 1, 2] = -translation[:, 1]
    ms = mat.clone()
    ms *= scale[:, None, None]
    ms[:, :, 2] = torch.sum(mat[:, :2, :2] * ms[:, :, 2].unsqueeze(-2), dim=-1)
    return ms


def warp_images(img, rot, scale, translation, white_background):
    grid = F.affine_grid(warp_matrix(rot, scale, translation), img.shape, align_corners=False)
    if white_background:
        img = img - 1
    out = F.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=False)
    if white_background:
        out = out + 1
    return out


def psnr(pred, target):
    """Per-image PSNR, inputs in [0,1] ([B,3,H,W] or [B,H,W,3])."""
    pred, target = pred.clamp(0, 1), target.clamp(0, 1)
    return (-10 * torch.log10((pred - target).square().mean(dim=[1, 2, 3]))).clamp(max=60)


def iou(alpha_pred, alpha_real):
    a, b = alpha_pred > 0.5, alpha_real > 0.5
    inter = (a & b).float().sum(dim=[-2, -1])
    union = (a | b).float().sum(dim=[-2, -1])
    return ((inter + 1e-6) / (union + 1e-6)).flatten()


def torgb_upsample_add(x, styles, weight, bias, previous_image):
    """x [B,Cin,R,R], styles [B,Cin] (affine(w) * weight_gain), weight [96,Cin,1,1], bias [96], previous_image
    [B,96,R/2,R/2] or None -> img [B,96,R,R]   (stylegan.py:132-133 modulation, 109 conv, 369-371 bias, 72-76 + 428-433)."""
    bs = x.shape[0]
    y = F.conv2d(x * styles.reshape(bs, -1, 1, 1), weight.reshape(weight.shape[0], -1, 1, 1), padding=0)
    y = y + bias.view(1, -1, 1, 1)
    if previous_image is None:
        return y
    h = torch.tensor([1., 3., 3., 1.], dtype=x.dtype, device=x.device)
    h = h[:, None] * h[None, :]
    h = h / h.sum()
    nc = previous_image.shape[1]
    up = F.conv_transpose2d(previous_image.flatten(0, 1).unsqueeze(1), h[None, None] * 4, padding=1, stride=2)
    up = up.view(bs, nc, up.shape[2], up.shape[3])
    return up + y
