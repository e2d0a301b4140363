"""A stand-in for the reference Generator's plane producer (the GPU box has no reference checkout).

Same attribute names and call conventions as models/generator.py:336-405 for everything the hot
path touches (mapping_network(.backbone.num_ws), synthesis_network, texture_mapper, decoder.net[0|2],
beta, alpha); the producers themselves are tiny so tests stay fast."""
import math

import torch
from torch import nn


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

    def __init__(self, res, gen):
        super().__init__()
        self.res = res
        self.basis = nn.Parameter(torch.randn(16, 96, res, res, generator=gen))
        self.proj = nn.Linear(512, 16)

    def forward(self, ws, **kw):
        coef = self.proj(ws.mean(dim=1))
        return torch.einsum('bk,kchw->bchw', coef, self.basis)


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
    def __init__(self, scene_range, attention_values=10, use_sdf=True, plane_res=64, seed=5, use_viewdir=False):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.scene_range = scene_range
        self.attention_values = attention_values
        self.use_sdf = use_sdf
        self.use_viewdir = use_viewdir
        self.use_encoder = False
        self.num_classes = None
        self.mapping_network = _Mapping(15 if attention_values > 0 else 14, gen)
        self.synthesis_network = _Synthesis(plane_res, gen)
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
