"""One-process-per-GPU replacements for the reference's ``nn.DataParallel`` use (run.py:560-644).

The reference scatters the batch along dim 0 to its GPUs, re-broadcasts all parameters on every
forward and reduces gradients to GPU 0 on every backward.  Here every rank holds a persistent
replica, takes its slice of the batch with :func:`shard_batch` (same dim-0 split as DP's scatter),
renders locally (no collective on the render path: rays are independent) and, when training, keeps
its gradients in :class:`GradientBuckets` - persistent flat fp32 buffers (128.7 MB for G, 115.7 MB
for D) whose buckets are all-reduced (RCCL over xGMI on the GPU box, gloo in the CPU tests)
asynchronously while backward is still running; :func:`allreduce_gradients` is the one-shot form.  Inversion needs no collective (per-image independent problems,
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


def shard_rows(height, rank=None, world_size=None, align=8):
    """[row0, row1) of an image `height` rows tall for this rank when ONE image is rendered by all ranks (SURVEY.md 8(e):
    "for a single huge image, shard pixel tiles and replicate the planes"; run.py:598-605 res_multiplier renders, or
    any batch smaller than the node).  Contiguous bands of whole `align`-row strips (the render kernel hands rays out in
    8 / 16 / 32-pixel blocks), as even as the strips allow; ranks beyond the strip count get an empty band."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    strips = -(-height // align)
    lo = (strips * rank) // world_size
    hi = (strips * (rank + 1)) // world_size
    return min(lo * align, height), min(hi * align, height)


def gather_rows(window, dim=1):
    """All-gathers the row bands of :func:`shard_rows` (tensors [B, rows_r, W, ...], rows along `dim`) into the full
    image on every rank.  Bands may differ in height (or be empty)."""
    rank, w = world()
    if w == 1:
        return window
    t = window.movedim(dim, 0).contiguous()
    full = gather_metrics(t)
    return full.movedim(0, dim).contiguous()


def allreduce_ray_setup(workspace, group=None):
    """Makes the batch-wide miss-fill of a row-sharded render the FULL image's (lib/nerf_utils.py:258-259: missed rays take
    min(near) / max(far) over the rays of the batch that hit the cube).  `workspace` is what ops.render_setup filled for
    this rank's row window; its first three 32-bit cells hold order-preserving keys of (min near, max far) over the
    window's hit rays and their count (csrc: raygen_kernel).  Max-reducing the two keys and summing the count over the
    ranks between the set-up and ops.render_fwd(rays_ready=True) gives every band the fill of the whole image, so a
    band is bit-identical to the same rows of the full render even for rays that are marched although they miss the
    exact cube (or with skip_missed_rays off).  Without it the identity holds only where no such ray is marched."""
    rank, w = world()
    if group is not None:
        w = dist.get_world_size(group)
    if w == 1:
        return
    cells = workspace[:12].view(torch.int32)
    v = cells.to(torch.int64) & 0xFFFFFFFF            # the cells are unsigned
    dist.all_reduce(v[:2], op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(v[2:], op=dist.ReduceOp.SUM, group=group)
    cells.copy_(torch.where(v >= 2 ** 31, v - 2 ** 32, v).to(torch.int32))


def join_ray_setup(device, group=None):
    """What a rank WITHOUT rows does while the others run :func:`allreduce_ray_setup` (row_window_sync): it enters the
    same reductions with the neutral element - zero-initialised cells are what the ray set-up starts from (csrc:
    reduce[0..2], atomicMax / atomicAdd on zeroed memory) - so that every rank of the group issues the same collectives.
    Pass it as ``render_image_rows(..., on_empty=lambda: join_ray_setup(device))`` when the bands are rendered with
    row_window_sync on."""
    allreduce_ray_setup(torch.zeros(16, dtype=torch.uint8, device=device), group)


def render_image_rows(render_rows, height, dim=1, on_empty=None):
    """One image over all ranks: `render_rows(row0, row1)` renders this rank's band (e.g. a render bound with
    ``make_render(..., row_window=(row0, row1 - row0))`` or ops.render_fwd(row_window=(row0, height))) and returns a tensor
    or a tuple of tensors with the rows along `dim`; the bands are gathered into full images on every rank.

    A rank whose band is EMPTY (more ranks than 8-row strips: a 32-row image on 8 ranks) does not call `render_rows` -
    the render kernels refuse a zero-row image - and contributes zero-row tensors instead; their shapes and dtypes come
    from a small metadata exchange with the ranks that did render, so every rank still enters the same collectives.
    If `render_rows` itself runs collectives (a render bound with row_window_sync reduces the miss-fill over the ranks
    between its ray set-up and its render kernel), the empty-band ranks have to enter those too: `on_empty()` is called
    on them instead of `render_rows` (``on_empty=lambda: join_ray_setup(device)``)."""
    r0, r1 = shard_rows(height)
    rank, w = world()
    out = render_rows(r0, r1) if r1 > r0 else None
    if r1 <= r0 and on_empty is not None and w > 1:
        on_empty()
    if w == 1:
        if out is None:
            raise ValueError('render_image_rows: the image has no rows')
        return tuple(out) if isinstance(out, (tuple, list)) else out
    single = out is not None and not isinstance(out, (tuple, list))
    items = None if out is None else ([out] if single else list(out))
    spec = None if items is None else dict(
        single=single, device=str(next(o.device for o in items if o is not None).type),
        items=[None if o is None else (tuple(o.shape), str(o.dtype).replace('torch.', '')) for o in items])
    specs = [None] * w
    dist.all_gather_object(specs, spec)
    have = next((sp for sp in specs if sp is not None), None)
    if have is None:
        raise ValueError('render_image_rows: no rank has rows to render (height %d)' % height)
    if items is None:
        dev = torch.device('cuda', torch.cuda.current_device()) if have['device'] == 'cuda' else torch.device(have['device'])
        items = []
        for it in have['items']:
            if it is None:
                items.append(None)
                continue
            shape = list(it[0])
            shape[dim] = 0
            items.append(torch.zeros(shape, dtype=getattr(torch, it[1]), device=dev))
    res = tuple(None if o is None else gather_rows(o, dim) for o in items)
    return res[0] if have['single'] else res


class GradientBuckets:
    """Persistent flat fp32 gradient storage for a replica's parameters, with the collective overlapped with backward.

    What it replaces: ``nn.DataParallel``'s per-backward reduce of every gradient to GPU 0 and per-forward re-broadcast
    of all parameters (run.py:636-644).  Here every rank keeps ONE set of flat buffers ("buckets",
    ``bucket_bytes`` each, filled in reverse parameter order = roughly the order in which backward produces gradients)
    and every ``param.grad`` is a VIEW into its bucket, so nothing is packed or copied back.  A post-accumulate hook per
    parameter counts arrivals; when a bucket is complete its collective is launched asynchronously (RCCL runs it on its
    own stream, over xGMI) while backward continues with the earlier layers.  ``finish()`` launches what is left (buckets
    containing parameters that received no gradient) and waits.  Buckets go out in the order in which backward completes
    them (the leftovers in index order): every rank runs the same graph (SPMD), so all ranks issue the same sequence.

    mode 'all_reduce'      one all-reduce per bucket;
    mode 'reduce_scatter'  reduce-scatter + all-gather per bucket (the two halves of a ring all-reduce issued
                           explicitly: on the fully connected xGMI mesh each is one direct exchange per peer).

    Use ``zero_grad()`` of this object instead of the optimiser's (which would detach the views).

    SEVERAL backward() calls per optimiser step (the reference accumulates: one ``loss.backward()`` per discriminator
    before ``optimizer_g.step()``, run.py:1044, and ``loss_real.backward()`` + ``loss_fake.backward()`` before
    ``optimizer_d.step()``, run.py:1110-1139): run all but the LAST of them inside ``with buckets.no_sync():`` -
    gradients then only accumulate in the flat buffers - and the last one outside, which launches the collectives::

        buckets.zero_grad()
        with buckets.no_sync():
            loss_a.backward()
        loss_b.backward()              # buckets go out as this backward completes them
        buckets.finish(); optimiser.step()

    A backward that reaches a bucket whose collective is already in flight would add local-only gradients to a buffer
    that has been (or is being) reduced - silently wrong and rank-divergent - so it raises instead.
    """

    def __init__(self, parameters, bucket_bytes=32 << 20, average=False, mode='all_reduce', overlap=True, group=None):
        if mode not in ('all_reduce', 'reduce_scatter'):
            raise ValueError("mode must be 'all_reduce' or 'reduce_scatter'")
        self.params = [p for p in parameters if p.requires_grad]
        if not self.params:
            raise ValueError('GradientBuckets: no parameter requires grad')
        self.average, self.mode, self.overlap, self.group = average, mode, overlap, group
        # a collective is issued whenever a process group exists - also with a single rank (bench.py --force-dist
        # runs the RCCL path on one GPU); without a process group the object only manages the flat storage
        self.active = dist.is_available() and dist.is_initialized()
        self.world_size = dist.get_world_size(group) if self.active else 1     # of THIS group: shard padding, averaging
        self._sync = True
        dev = self.params[0].device
        per = max(1, int(bucket_bytes) // 4)
        self.buckets = []          # dict(flat, params, offsets, pending, shard)
        cur, cur_n = [], 0
        for p in reversed(self.params):
            if p.dtype != torch.float32 or p.device != dev:
                raise TypeError('GradientBuckets: fp32 parameters on one device only')
            if cur and cur_n + p.numel() > per:
                self._close(cur, cur_n, dev)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        self._close(cur, cur_n, dev)
        self._bucket_of = {}
        self._ptr = {}
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for p in b['params']:
                self._bucket_of[id(p)] = bi
                self._hooks.append(p.register_post_accumulate_grad_hook(self._arrived))
        self.nbytes = sum(b['numel'] for b in self.buckets) * 4
        self.zero_grad()

    def _close(self, params, n, dev):
        w = max(self.world_size, 1)
        padded = -(-n // w) * w                          # reduce-scatter needs equal shards
        flat = torch.zeros(padded, dtype=torch.float32, device=dev)
        offs, o = [], 0
        for p in params:
            offs.append(o)
            o += p.numel()
        shard = torch.empty(padded // w, dtype=torch.float32, device=dev) if self.mode == 'reduce_scatter' else None
        self.buckets.append(dict(flat=flat, params=params, offsets=offs, numel=n, shard=shard))

    def zero_grad(self):
        """Zeroes the flat buffers and (re-)attaches every param.grad as a view of its bucket."""
        for b in self.buckets:
            b['flat'].zero_()
            for p, o in zip(b['params'], b['offsets']):
                v = b['flat'][o:o + p.numel()].view_as(p)
                if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                    p.grad = v
                self._ptr[id(p)] = v.data_ptr()
        self._arrivals = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self.launched_in_backward = 0

    def no_sync(self):
        """Context manager for every backward of a step except the last one: gradients accumulate, nothing is launched
        and arrivals are not counted (see the class docstring)."""
        buckets = self

        class _NoSync:
            def __enter__(self):
                if any(buckets._launched):
                    raise RuntimeError('GradientBuckets.no_sync(): collectives of this step are already in flight; '
                                       'accumulating backwards must come BEFORE the synchronising one')
                buckets._sync = False

            def __exit__(self, *exc):
                buckets._sync = True
                return False
        return _NoSync()

    def _launch(self, bi):
        b = self.buckets[bi]
        if not self.active:
            return
        if self.mode == 'all_reduce':
            self._handles.append((bi, dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
        else:
            self._handles.append((bi, dist.reduce_scatter_tensor(b['shard'], b['flat'], op=dist.ReduceOp.SUM,
                                                                 group=self.group, async_op=True)))

    def _arrived(self, p):
        bi = self._bucket_of[id(p)]
        if p.grad.data_ptr() != self._ptr[id(p)]:
            raise RuntimeError('GradientBuckets: a .grad was replaced (use GradientBuckets.zero_grad(), not the '
                               "optimiser's zero_grad(set_to_none=True))")
        if self._launched[bi] and self.active:
            raise RuntimeError('GradientBuckets: a gradient arrived in a bucket whose collective was already launched '
                               '(a second backward() in this step?).  Run every backward but the last inside '
                               '`with buckets.no_sync():`, or call zero_grad() between steps')
        if not self._sync or not self.overlap:
            return                       # (without overlap nothing is launched before finish(): any number of backwards)
        self._arrivals[bi] += 1
        if self._arrivals[bi] > len(self.buckets[bi]['params']):
            raise RuntimeError('GradientBuckets: a parameter received two gradients in one synchronising backward pass '
                               '(two backward() calls without no_sync()?)')
        if self._arrivals[bi] == len(self.buckets[bi]['params']) and not self._launched[bi]:
            self._launch(bi)
            self._launched[bi] = True
            self.launched_in_backward += 1

    def finish(self):
        """Launches the collectives not yet issued, waits for all of them, applies the 1/world_size of `average`.
        Returns the number of gradient bytes reduced (0 without a process group)."""
        if not self.active:
            return 0
        for bi in range(len(self.buckets)):
            if not self._launched[bi]:
                self._launch(bi)
                self._launched[bi] = True
        gathers = []
        for bi, h in self._handles:
            h.wait()
            b = self.buckets[bi]
            if self.mode == 'reduce_scatter':
                if self.average and self.world_size > 1:
                    b['shard'] /= self.world_size
                gathers.append(dist.all_gather_into_tensor(b['flat'], b['shard'], group=self.group, async_op=True))
            elif self.average and self.world_size > 1:
                b['flat'] /= self.world_size
        for h in gathers:
            h.wait()
        self._handles = []
        return self.nbytes

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


_FLAT_CACHE = {}
_FLAT_CACHE_MAX = 4


def allreduce_gradients(parameters, average=False, group=None):
    """Sums (or averages) .grad of the given parameters across the ranks of `group`, one shot after backward (no
    overlap; for the persistent, overlapped form use :class:`GradientBuckets`).  The gradients go through a persistent
    flat staging buffer per (device, size) - generator and discriminator steps alternate every iteration, so both
    sizes stay cached (at most four buffers are kept, oldest dropped first)."""
    params = [p for p in parameters if p.grad is not None]
    if not (dist.is_available() and dist.is_initialized()) or not params:
        return 0
    w = dist.get_world_size(group)
    if w == 1:
        return 0
    n = sum(p.grad.numel() for p in params)
    key = (params[0].grad.device, n)
    flat = _FLAT_CACHE.pop(key, None)
    if flat is None:
        while len(_FLAT_CACHE) >= _FLAT_CACHE_MAX:
            _FLAT_CACHE.pop(next(iter(_FLAT_CACHE)))
        flat = torch.empty(n, dtype=torch.float32, device=key[0])
    _FLAT_CACHE[key] = flat                    # (re-)inserted last: most recently used
    views, off = [], 0
    for p in params:
        k = p.grad.numel()
        views.append(flat[off:off + k].view_as(p.grad))
        off += k
    torch._foreach_copy_(views, [p.grad for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= w
    torch._foreach_copy_([p.grad for p in params], views)
    return n * 4


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
