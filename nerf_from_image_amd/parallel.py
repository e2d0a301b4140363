"""One-process-per-GPU replacements for the reference's ``nn.DataParallel`` use (run.py:560-644).

The reference scatters the batch along dim 0 to its GPUs, re-broadcasts all parameters on every
forward and reduces gradients to GPU 0 on every backward.  Here every rank holds a persistent
replica, takes its slice of the batch with :func:`shard_batch` (same dim-0 split as DP's scatter),
renders locally (no collective on the render path: rays are independent) and, when training, calls
:func:`allreduce_gradients` once per optimiser step: the gradients are packed into ONE flat fp32
buffer (128.7 MB for G, 115.7 MB for D) and summed with a single all-reduce (RCCL over xGMI on the
GPU box, gloo in the CPU tests).  Inversion needs no collective (per-image independent problems,
run.py:2232-2241); :func:`gather_metrics` collects per-image results for the final report.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n, rank=None, world_size=None):
    """[start, end) of this rank's slice of a batch of n (torch.chunk semantics, like DP's scatter)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    per = -(-n // world_size)
    start = min(rank * per, n)
    return start, min(start + per, n)


def shard_batch(*tensors, rank=None, world_size=None):
    """Slices every tensor (or None) along dim 0 for this rank."""
    out = []
    for t in tensors:
        if t is None:
            out.append(None)
            continue
        a, b = shard_range(t.shape[0], rank, world_size)
        out.append(t[a:b])
    return out[0] if len(out) == 1 else tuple(out)


def allreduce_gradients(parameters, average=False):
    """Sums (or averages) .grad of the given parameters across ranks with ONE all-reduce."""
    params = [p for p in parameters if p.grad is not None]
    rank, w = world()
    if w == 1 or not params:
        return 0
    flat = torch.cat([p.grad.reshape(-1).float() for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= w
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
    return flat.numel() * 4


def allreduce_scalar_mean(value, device=None):
    """Mean over ranks of a scalar statistic (regulariser means, ppl_running_avg, run.py:1034-1038)."""
    rank, w = world()
    t = torch.as_tensor(float(value), dtype=torch.float64, device=device)
    if w > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= w
    return float(t)


def gather_metrics(t):
    """All-gathers a per-image tensor [b_local, ...] into [b_global, ...] on every rank."""
    rank, w = world()
    if w == 1:
        return t
    sizes = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(w)]
    dist.all_gather(sizes, torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device))
    m = int(max(s.item() for s in sizes))
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    bufs = [torch.zeros_like(pad) for _ in range(w)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:int(s.item())] for b, s in zip(bufs, sizes)])


class ParallelModel(torch.nn.Module):
    """One-process-per-GPU counterpart of run.py's ``ParallelModel`` (560-617): same constructor
    arguments and the same ``forward`` signature/dispatch, with ``render`` being the HIP drop-in and
    ``depth_samples_per_ray`` passed in instead of read from a script global (run.py:512-514).

    Each rank builds ONE instance around its persistent replicas; callers shard the batch with
    :func:`shard_batch` instead of relying on DataParallel's scatter."""

    def __init__(self, resolution, model=None, model_ema=None, lpips_net=None, render=None,
                 depth_samples_per_ray=64):
        super().__init__()
        self.resolution = resolution
        self.model = model
        self.model_ema = model_ema
        self.lpips_net = lpips_net
        self._render = render
        self.depth_samples_per_ray = depth_samples_per_ray

    def forward(self, tform_cam2world, focal, center, bbox, c, use_ema=False, ray_multiplier=1, res_multiplier=1,
                pretrain_sdf=False, compute_normals=False, compute_semantics=False, compute_coords=False,
                encoder_output=False, closure=None, closure_params=None, extra_model_outputs=[],
                extra_model_inputs={}, force_no_cam_grad=False):
        model_to_use = self.model_ema if use_ema else self.model
        if pretrain_sdf:
            return model_to_use(None, c, request_model_outputs=['sdf_distance_loss', 'sdf_eikonal_loss'])
        if encoder_output:
            return model_to_use.emb(c)
        render = self._render
        if render is None:
            from . import render as nfi_render
            render = nfi_render.render
        res = int(self.resolution * res_multiplier)
        output = render(model_to_use, res, res, tform_cam2world, focal, center, bbox, c,
                        self.depth_samples_per_ray * ray_multiplier, compute_normals=compute_normals,
                        compute_semantics=compute_semantics, compute_coords=compute_coords,
                        extra_model_outputs=extra_model_outputs, extra_model_inputs=extra_model_inputs,
                        force_no_cam_grad=force_no_cam_grad)
        if closure is not None:
            # RGB, alpha, semantics, extra outputs - as run.py:612-615
            return closure(self, output[0], output[2], output[4], output[-1], **closure_params)
        return output
