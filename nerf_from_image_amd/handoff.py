"""Plane-producer hand-off without a transposition.

The reference's synthesis network ends with ``img = upsample2d(img) ; y = torgb(x, w) ; img = img.add_(y)``
(models/stylegan.py:424-433, the last ``SynthesisBlock``) and hands the NCHW result to the renderer, which in round 1
cost one NCHW -> channel-last pass per render (and one back per backward).  ``fuse_last_block`` replaces the tail of
that block by ONE HIP kernel (``nfi_torgb_texels_fwd``: 1x1 modulated conv on MFMA + bias + bilinear up-filter + add)
whose output is a ``[B,96,R,R]`` tensor in channels-last memory format - i.e. exactly the interleaved texel image the
field kernels read, while every PyTorch consumer of ``planes`` still sees the shape it expects.  Parameters, buffers
and ``state_dict`` keys are untouched; the block's two 3x3 modulated convolutions stay the reference's modules.

First-order only: the fused node has a HIP backward but no double backward, so a forward that asks for the
``path_length`` output (``torch.autograd.grad(planes * noise, ws, create_graph=True)``, generator.py:484-499, which
is then differentiated AGAIN) runs the block's original PyTorch tail for that call - ``with unfused(net):`` in
``generator.hip_forward`` / ``wrapped_forward`` - instead of back-propagating an incomplete second-order gradient.

    import nerf_from_image_amd.handoff as nfi_handoff
    nfi_handoff.fuse_last_block(model.synthesis_network)          # or attach(model, fused_handoff=True)
"""
import types

import torch

from . import ops
from .autograd import differentiable


def torgb_upsample_add(x, styles, weight, bias, previous_image):
    """Differentiable fused tail: x [B,Cin,R,R], styles [B,Cin] (already times weight_gain), weight [96,Cin,1,1] or
    [96,Cin], bias [96], previous_image [B,96,R/2,R/2] or None -> [B,96,R,R] channels-last."""
    if (not x.is_cuda) or weight.shape[0] != 96 or x.shape[1] % 16 or x.shape[1] > 256 or x.shape[-1] % 8:
        raise RuntimeError('fused hand-off: needs a GPU tensor, 96 image channels, a multiple of 16 (<= 256) feature '
                           'channels and a resolution that is a multiple of 8; got x %s' % (tuple(x.shape),))
    w2 = weight.reshape(weight.shape[0], -1)
    has_prev = previous_image is not None

    def fwd(a_x, a_s, a_w, a_b, *a_prev):
        return ops.torgb_texels(a_x, a_s, a_w, a_b, a_prev[0] if has_prev else None)

    def bwd(inputs, out_meta, grads, needs):
        a_x, a_s, a_w = inputs[0], inputs[1], inputs[2]
        prev = inputs[4] if has_prev else None
        g = ops.torgb_texels_bwd(grads[0], a_x, a_s, a_w, prev, want_weight=bool(needs[2] or needs[3]),
                                 want_prev=has_prev and bool(needs[4]))
        out = (g['g_x'], g['g_styles'], g.get('g_weight'), g.get('g_bias'))
        return out + ((g.get('g_previous_image'),) if has_prev else ())
    args = (x, styles, w2, bias) + ((previous_image,) if has_prev else ())
    return differentiable('torgb_texels', fwd, *args, bwd=bwd)


def fused_block_forward(self, x, img, ws, **layer_kwargs):
    """SynthesisBlock.forward (models/stylegan.py:416-435) with its tail on the HIP kernel."""
    w_iter = iter(ws.unbind(dim=1))
    if self.in_channels == 0:
        x = self.const.unsqueeze(0).repeat([ws.shape[0], 1, 1, 1])
    else:
        x = self.conv0(x, next(w_iter), **layer_kwargs)
    x = self.conv1(x, next(w_iter), **layer_kwargs)
    torgb = self.torgb
    styles = torgb.affine(next(w_iter)) * torgb.weight_gain               # OutputLayer.forward, stylegan.py:365
    img = torgb_upsample_add(x, styles, torgb.weight, torgb.bias, img)
    return x, img


def last_block(synthesis_network):
    res = getattr(synthesis_network, 'img_resolution', None)
    blk = getattr(synthesis_network, 'b%d' % res, None) if res is not None else None
    if blk is None or not all(hasattr(blk, a) for a in ('conv1', 'torgb', 'in_channels')):
        raise AttributeError('fuse_last_block: the synthesis network has no StyleGAN2-style last block b<img_resolution>')
    return blk


def fuse_last_block(synthesis_network):
    """Swaps the forward of the last synthesis block for ``fused_block_forward``.  Returns the block."""
    blk = last_block(synthesis_network)
    if not hasattr(blk, '_nfi_original_forward'):
        blk._nfi_original_forward = blk.forward
        blk.forward = types.MethodType(fused_block_forward, blk)
    return blk


class unfused:
    """``with unfused(net):`` - the last block's original forward for the duration of the block (no-op when the block
    is not fused)."""

    def __init__(self, synthesis_network):
        self.net = synthesis_network
        self.blk = None

    def __enter__(self):
        try:
            blk = last_block(self.net)
        except AttributeError:
            return self
        if hasattr(blk, '_nfi_original_forward'):
            self.blk = blk
            self.fused = blk.forward
            blk.forward = blk._nfi_original_forward
        return self

    def __exit__(self, *exc):
        if self.blk is not None:
            self.blk.forward = self.fused
        return False


def unfuse_last_block(synthesis_network):
    blk = last_block(synthesis_network)
    if hasattr(blk, '_nfi_original_forward'):
        blk.forward = blk._nfi_original_forward
        del blk._nfi_original_forward
    return blk
