"""A stand-in for the reference Generator's plane producer (the GPU box has no reference checkout).

Same attribute names and call conventions as models/generator.py:336-405 for everything the hot
path touches (mapping_network(.backbone.num_ws), synthesis_network, texture_mapper, decoder.net[0|2],
beta, alpha); the producers themselves are tiny so tests stay fast."""
import math
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
from basis_mix import basis_mix  # noqa: E402


class _Lin(nn.Module):
    def __init__(self, i, o, gen):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(o, i, generator=gen))
        self.bias = nn.Parameter(0.3 * torch.randn(o, generator=gen))


class _Decoder(nn.Module):
    def __init__(self, n_out, gen):
        super().__init__()
        self.net = nn.Sequential(_Lin(32, 64, gen), nn.Identity(), _Lin(64, n_out, gen))


class _Backbone(nn.Module):
    def __init__(self, num_ws):
        super().__init__()
        self.num_ws = num_ws


class _Mapping(nn.Module):
    def __init__(self, num_ws, gen):
        super().__init__()
        self.backbone = _Backbone(num_ws)
        self.lin = nn.Linear(512, 512)

    def forward(self, z, c=None):
        return self.lin(z).unsqueeze(1).expand(-1, self.backbone.num_ws, -1).contiguous()


class _Synthesis(nn.Module):
    """ws[:, :14] -> [B,96,R,R]: a fixed smooth basis modulated by the latents."""

    def __init__(self, res, gen, n_basis=16):
        super().__init__()
        self.res = res
        self.basis = nn.Parameter(torch.randn(16, 96, res, res, generator=gen)[:n_basis].clone())
        self.proj = nn.Linear(512, n_basis)

    def forward(self, ws, **kw):
        coef = self.proj(ws.mean(dim=1))
        return basis_mix(coef, self.basis)


class _Texture(nn.Module):
    def __init__(self, A):
        super().__init__()
        self.A = A
        self.lin = nn.Linear(512, A * 3)

    def forward(self, w):
        return torch.sigmoid(self.lin(w).view(-1, self.A, 3)) * 2.004 - 1.002


class _ViewDirMapper(nn.Module):
    """Shape-compatible stand-in of ViewDirectionMapper (models/generator.py:189-253): a per-ray MLP whose last
    hidden layer is `fc6` (its output is the per-ray feature) and whose `output` is the Linear(32, A or 3) the
    closure applies per sample (raw EqualizedLinear parameters, gain 1/sqrt(32))."""

    def __init__(self, n_out, gen):
        super().__init__()
        self.fc0 = nn.Linear(3, 64)
        self.fc6 = nn.Linear(64, 32)
        self.output = _Lin(32, n_out, gen)

    def forward(self, viewdir):
        return self.fc6(torch.nn.functional.leaky_relu(self.fc0(viewdir), 0.2))


class StandInGenerator(nn.Module):
    def __init__(self, scene_range, attention_values=10, use_sdf=True, plane_res=64, seed=5, use_viewdir=False, n_basis=16):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.scene_range = scene_range
        self.attention_values = attention_values
        self.use_sdf = use_sdf
        self.use_viewdir = use_viewdir
        self.use_encoder = False
        self.num_classes = None
        self.mapping_network = _Mapping(15 if attention_values > 0 else 14, gen)
        self.synthesis_network = _Synthesis(plane_res, gen, n_basis)
        self.decoder = _Decoder(33 if use_viewdir else (1 + attention_values if attention_values > 0 else 4), gen)
        if use_viewdir:
            self.viewdir_mapper = _ViewDirMapper(attention_values if attention_values > 0 else 3, gen)
        if attention_values > 0:
            self.texture_mapper = _Texture(attention_values)
        if use_sdf:
            self.beta = nn.Parameter(torch.FloatTensor([0.1]))
            self.alpha = nn.Parameter(torch.FloatTensor([0.05]))

    def planes_and_values(self, c):
        """What hip_forward will hand to the kernels, computed the plain way (for the oracle)."""
        ws = self.mapping_network(c) if c.dim() == 2 else c
        att = self.texture_mapper(ws[:, 14]) if self.attention_values > 0 else None
        planes = self.synthesis_network(ws[:, :14])
        return planes.view(c.shape[0], 3, 32, planes.shape[-2], planes.shape[-1]), att


def look_at_cameras(n, radius, gen):
    v = torch.randn(n, 3, generator=gen)
    eye = radius * v / v.norm(dim=-1, keepdim=True)
    fwd = -eye / eye.norm(dim=-1, keepdim=True)
    up = torch.tensor([0., 0., 1.]).expand(n, 3)
    right = torch.cross(fwd, up, dim=-1)
    right = right / right.norm(dim=-1, keepdim=True)
    tup = torch.cross(right, fwd, dim=-1)
    cam = torch.eye(4).repeat(n, 1, 1)
    cam[:, :3, 0] = right
    cam[:, :3, 1] = tup
    cam[:, :3, 2] = -fwd
    cam[:, :3, 3] = eye
    return cam


# ------------------------------------------------------------------------------------------------
# a synthesis network with the reference's LAST-BLOCK structure (models/stylegan.py:383-435), small
# ------------------------------------------------------------------------------------------------
class _Affine(nn.Module):
    def __init__(self, w_dim, out):
        super().__init__()
        self.lin = nn.Linear(w_dim, out)
        nn.init.ones_(self.lin.bias)

    def forward(self, w):
        return self.lin(w)


class _ModConv(nn.Module):
    """Stand-in for SynthesisLayer: style-modulated 3x3 conv (+ optional 2x nearest upsampling), leaky ReLU."""

    def __init__(self, cin, cout, w_dim, up, gen):
        super().__init__()
        self.up = up
        self.affine = _Affine(w_dim, cin)
        self.weight = nn.Parameter(torch.randn(cout, cin, 3, 3, generator=gen) / math.sqrt(cin * 9))
        self.bias = nn.Parameter(torch.zeros(cout))

    def forward(self, x, w, **kw):
        x = x * self.affine(w)[:, :, None, None]
        if self.up:
            x = torch.nn.functional.interpolate(x, scale_factor=2, mode='nearest')
        return torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x, self.weight, self.bias, padding=1), 0.2)


class _OutputLayer(nn.Module):
    """Same attributes as the reference OutputLayer (stylegan.py:351-372): affine, weight [96,Cin,1,1], bias, weight_gain."""

    def __init__(self, cin, w_dim, gen):
        super().__init__()
        self.affine = _Affine(w_dim, cin)
        self.weight = nn.Parameter(torch.randn(96, cin, 1, 1, generator=gen))
        self.bias = nn.Parameter(0.1 * torch.randn(96, generator=gen))
        self.weight_gain = 1 / math.sqrt(cin)

    def forward(self, x, w):
        styles = self.affine(w) * self.weight_gain
        y = torch.nn.functional.conv2d(x * styles[:, :, None, None], self.weight)
        return y + self.bias.view(1, -1, 1, 1)


class _Block(nn.Module):
    def __init__(self, cin, cout, w_dim, res, gen):
        super().__init__()
        self.in_channels, self.resolution = cin, res
        f = torch.tensor([1., 3., 3., 1.])
        f = f[:, None] * f[None, :]
        self.register_buffer('resample_filter', f / f.sum())
        if cin == 0:
            self.const = nn.Parameter(torch.randn(cout, res, res, generator=gen))
        else:
            self.conv0 = _ModConv(cin, cout, w_dim, True, gen)
        self.conv1 = _ModConv(cout, cout, w_dim, False, gen)
        self.torgb = _OutputLayer(cout, w_dim, gen)
        self.num_conv = 1 if cin == 0 else 2

    def forward(self, x, img, ws, **kw):
        """The reference block (stylegan.py:416-435) in plain tensor ops."""
        w_iter = iter(ws.unbind(dim=1))
        if self.in_channels == 0:
            x = self.const.unsqueeze(0).repeat([ws.shape[0], 1, 1, 1])
        else:
            x = self.conv0(x, next(w_iter))
        x = self.conv1(x, next(w_iter))
        if img is not None:
            nc = img.shape[1]
            up = torch.nn.functional.conv_transpose2d(img.flatten(0, 1).unsqueeze(1), self.resample_filter[None, None] * 4,
                                                      padding=1, stride=2)
            img = up.view(img.shape[0], nc, up.shape[2], up.shape[3])
        y = self.torgb(x, next(w_iter))
        img = img + y if img is not None else y
        return x, img


class StyleLikeSynthesis(nn.Module):
    """Two blocks (res/2 with a learned constant, res = the LAST block) driven like SynthesisNetwork.forward
    (stylegan.py:477-492): ws [B,4,w_dim]: b(res/2) uses ws[0..1] (conv1, torgb), b(res) uses ws[1..3]."""

    def __init__(self, res, channels=32, w_dim=512, seed=3):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.img_resolution, self.img_channels, self.w_dim = res, 96, w_dim
        self.block_resolutions = [res // 2, res]
        setattr(self, 'b%d' % (res // 2), _Block(0, channels, w_dim, res // 2, gen))
        setattr(self, 'b%d' % res, _Block(channels, channels, w_dim, res, gen))

    def forward(self, ws, **kw):
        ws = ws[:, :4]
        x, img = getattr(self, 'b%d' % (self.img_resolution // 2))(None, None, ws[:, 0:2])
        x, img = getattr(self, 'b%d' % self.img_resolution)(x, img, ws[:, 1:4])
        return img
