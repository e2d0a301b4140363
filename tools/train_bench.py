"""Generator-side training step in BASELINE config 4's geometry, one process per GPU, with the RCCL gradient
all-reduce the north star names (run.py:636-644 + the G step run.py:980-1073 are what it stands in for).

Per step and rank: render 4 images at 128x128 with 64+64 samples through the DIFFERENTIABLE HIP path (orthographic
camera, scene_range 2.0, black background, image + alpha loss as with --supervise_alpha), the SDF regulariser branch
(eikonal + distance, every G step in cfg4), backward through the HIP backward kernels into the plane producer /
decoder / beta / alpha, all-reduce of ONE generator-sized fp32 gradient set (32.2 M parameters = 128.7 MB, SURVEY.md
8(e)) through nerf_from_image_amd.parallel.GradientBuckets (buckets launched asynchronously while backward is still
running), Adam step.  The discriminator, the data pipeline and the StyleGAN2 synthesis network are outside the hot
path: the plane producer here is a small latent-modulated basis padded with tensors of the reference
generator's own parameter list (tools/generator_param_sizes.json: 129 tensors, 1 ... 2 359 296 elements,
models/stylegan.py:438-490), so that the collective has the real size AND the real bucket layout.

Used by `bench.py --mode train` (and runnable alone under torch.distributed.run)."""
import math
import time

import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from basis_mix import basis_mix  # noqa: E402

G_PARAMS = 32_175_000          # reference Generator: 128.7 MB of fp32 gradients


class _Lin(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(o, i))
        self.bias = nn.Parameter(0.3 * torch.randn(o))


class _Decoder(nn.Module):
    def __init__(self, n_out):
        super().__init__()
        self.net = nn.Sequential(_Lin(32, 64), nn.Identity(), _Lin(64, n_out))


class _Backbone(nn.Module):
    num_ws = 15


class _Mapping(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = _Backbone()
        self.lin = nn.Linear(512, 512)

    def forward(self, z, c=None):
        return self.lin(z).unsqueeze(1).expand(-1, 15, -1).contiguous()


def reference_parameter_shapes():
    """Shapes of the reference Generator's parameters in registration order (tools/generator_param_sizes.json, written
    from the live class by tools/make_generator_param_sizes.py: 129 tensors, 32.16 M elements)."""
    import json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'generator_param_sizes.json')
    return [tuple(p['shape']) for p in json.load(open(path))['parameters']]


class _Synthesis(nn.Module):
    """ws[:, :14] -> [B,96,R,R]: K smooth basis images modulated by the latents, plus padding parameters with the
    reference generator's OWN tensor list (one tensor per reference parameter, same sizes, same order), so that the
    gradient buckets - their number, sizes and the per-tensor gradient writes - are the reference's; they receive zero
    gradients and ride along in the all-reduce."""

    def __init__(self, res, k, pad_shapes):
        super().__init__()
        self.res = res
        self.basis = nn.Parameter(torch.randn(k, 96, 16, 16))          # low resolution: the stand-in's own parameters stay
        self.proj = nn.Linear(512, k)                                  # small next to the reference's tensor list
        self.padding = nn.ParameterList([nn.Parameter(torch.zeros(s if len(s) else (1,))) for s in pad_shapes])

    def forward(self, ws, **kw):
        coef = self.proj(ws.mean(dim=1))
        basis = torch.nn.functional.interpolate(self.basis, size=(self.res, self.res), mode='bilinear', align_corners=True)
        out = basis_mix(coef, basis)
        if len(self.padding):
            out = out + 0.0 * torch.stack([p.reshape(-1)[0] for p in self.padding]).sum()
        return out


class _Texture(nn.Module):
    def __init__(self, a):
        super().__init__()
        self.a = a
        self.lin = nn.Linear(512, a * 3)

    def forward(self, w):
        return torch.sigmoid(self.lin(w).view(-1, self.a, 3)) * 2.004 - 1.002


class PlaneProducer(nn.Module):
    """Bare container with the attributes nerf_from_image_amd.generator.attach() needs (REQUIRED_ATTRS)."""

    def __init__(self, scene_range, plane_res=256, k=4, total_params=G_PARAMS):
        super().__init__()
        self.scene_range, self.attention_values, self.use_sdf = scene_range, 10, True
        self.use_viewdir = self.use_encoder = False
        self.num_classes = None
        self.mapping_network = _Mapping()
        self.decoder = _Decoder(11)
        self.texture_mapper = _Texture(10)
        self.beta = nn.Parameter(torch.tensor([0.1]))
        self.alpha = nn.Parameter(torch.tensor([0.05]))
        self.synthesis_network = _Synthesis(plane_res, k, [])
        # pad with the reference's tensors, largest-first skipping as many elements as the stand-in's own modules hold
        have = sum(p.numel() for p in self.parameters())
        shapes = reference_parameter_shapes()
        skip = []
        for i in sorted(range(len(shapes)), key=lambda j: -int(torch.tensor(shapes[j] or (1,)).prod())):
            n = int(torch.tensor(shapes[i] or (1,)).prod())
            if n <= have:
                skip.append(i)
                have -= n
        self.synthesis_network = _Synthesis(plane_res, k, [s for i, s in enumerate(shapes) if i not in set(skip)])


def run(dev, steps, warmup, batch=4, res=128, samples=64, bucket_mb=32, reduce_mode='all_reduce', overlap=True,
        use_dist=False, seed=0, texels='fp32'):
    """Returns a dict of timings (this rank) - the caller aggregates over ranks."""
    import types
    import torch.distributed as dist
    import nerf_from_image_amd.generator as nfi_gen
    import nerf_from_image_amd.render as nfi_render
    from nerf_from_image_amd.parallel import GradientBuckets
    rank = dist.get_rank() if use_dist else 0
    torch.manual_seed(seed)                                   # identical replicas on every rank
    scene_range = 2.0
    model = PlaneProducer(scene_range).to(dev).train()
    from nerf_from_image_amd import ops
    # texel storage the kernels gather from (arithmetic and the plane gradient stay fp32; 16-bit storage: the gradient is
    # taken w.r.t. the rounded planes, straight through)
    nfi_gen.attach(model, texel_dtype={'fp32': ops.TEXEL_F32, 'fp16': ops.TEXEL_F16, 'bf16': ops.TEXEL_BF16}[texels])
    g = torch.Generator().manual_seed(1000 + rank)            # every rank its own images
    v = torch.randn(batch, 3, generator=g)
    eye = 3.0 * v / v.norm(dim=-1, keepdim=True)
    fwd = -eye / eye.norm(dim=-1, keepdim=True)
    right = torch.cross(fwd, torch.tensor([0., 0., 1.]).expand(batch, 3), dim=-1)
    right = right / right.norm(dim=-1, keepdim=True)
    cam = torch.eye(4).repeat(batch, 1, 1)
    cam[:, :3, 0], cam[:, :3, 1], cam[:, :3, 2], cam[:, :3, 3] = right, torch.cross(right, fwd, dim=-1), -fwd, eye
    cam = cam.to(dev)
    z = torch.randn(batch, 512, generator=g).to(dev)
    target_rgb = (torch.rand(batch, res, res, 3, generator=g) * 2 - 1).to(dev)
    target_mask = (torch.rand(batch, res, res, generator=g) > 0.5).float().to(dev)
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=True)
    render = nfi_render.make_render(cfg, {'scene_range': scene_range, 'white_background': False},
                                    strict_near_far=False)   # no host synchronisation inside the step
    params = [p for p in model.parameters() if p.requires_grad]
    n_params = sum(p.numel() for p in params)
    buckets = GradientBuckets(params, bucket_bytes=bucket_mb << 20, average=True, mode=reduce_mode, overlap=overlap)
    opt = torch.optim.Adam(params, lr=0.0025, betas=(0.0, 0.99), fused=True)
    reg = ['sdf_eikonal_loss', 'sdf_distance_loss']
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t_fb, t_red, t_opt = [], [], []

    def step(timed):
        buckets.zero_grad()
        ev[0].record()
        rgb, _, mask, _, _, extra = render(model, res, res, cam, None, None, None, z, samples, extra_model_outputs=reg)
        loss = ((rgb - target_rgb) ** 2).mean() + ((mask - target_mask) ** 2).mean() + \
            0.1 * extra['sdf_eikonal_loss'].mean() + extra['sdf_distance_loss'].mean()
        loss.backward()
        ev[1].record()
        buckets.finish()                                      # waits on this stream for the collectives' stream
        ev[2].record()
        opt.step()
        with torch.no_grad():
            model.beta.clamp_(min=1e-3)                       # run.py:1069-1071
            model.alpha.clamp_(min=1e-3)
        ev[3].record()
        if timed:
            torch.cuda.synchronize()
            t_fb.append(ev[0].elapsed_time(ev[1]))
            t_red.append(ev[1].elapsed_time(ev[2]))
            t_opt.append(ev[2].elapsed_time(ev[3]))
        return loss

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        loss = step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step(False)
    fence()
    elapsed = time.perf_counter() - t0
    for _ in range(min(steps, 10)):                           # untimed extra steps with a sync per step: the breakdown
        loss = step(True)

    in_backward = buckets.launched_in_backward
    # the collective alone (no backward to hide behind): all buckets, launched back to back
    alone = []
    if use_dist:
        for i in range(5):
            buckets.zero_grad()
            fence()
            t1 = time.perf_counter()
            buckets.finish()
            torch.cuda.synchronize()
            if i > 0:
                alone.append((time.perf_counter() - t1) * 1e3)
    med = lambda xs: sorted(xs)[len(xs) // 2] if xs else None
    return dict(elapsed_s=elapsed, ms_per_step=elapsed / steps * 1e3, loss=float(loss.detach()),
                fwd_bwd_ms=med(t_fb), allreduce_exposed_ms=med(t_red), optimiser_ms=med(t_opt),
                allreduce_alone_ms=med(alone), gradient_bytes=buckets.nbytes, n_params=n_params,
                n_buckets=len(buckets.buckets), buckets_launched_in_backward=in_backward,
                rays_per_step=batch * res * res)
