"""Autograd glue between the HIP kernels and PyTorch.

Every host-level op goes through :func:`differentiable`: the forward runs the HIP kernel(s); if
any input requires grad, the call is recorded as ONE autograd node whose backward is the ``bwd``
closure handed in by the op (it launches the HIP backward kernels).  An op without a backward
fails loudly when a gradient is actually requested; nothing ever falls back to ATen.  The nodes are
first-order only: ``torch.autograd.grad(..., create_graph=True)`` THROUGH a node raises (the one
second-order path of the reference that touches the renderer's neighbourhood, the path-length
regulariser of the plane producer, is kept off the fused hand-off node: generator.hip_forward).

Lifetime: the node keeps its INPUTS through ``save_for_backward`` and, of its outputs, only their
shape/dtype/device (:class:`OutputMeta`).  Keeping an output tensor on ``ctx`` would close the
cycle output -> grad_fn -> ctx -> output, which the reference-counting of autograd never frees:
every step would leak its per-sample tensors.
"""
import torch


class OutputMeta:
    """What a backward closure may know about a forward output without keeping it alive."""
    __slots__ = ('shape', 'dtype', 'device')

    def __init__(self, t):
        self.shape, self.dtype, self.device = tuple(t.shape), t.dtype, t.device

    def zeros(self):
        return torch.zeros(self.shape, dtype=self.dtype, device=self.device)


class _HipNode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, name, fn, bwd, nondiff, *tensors):
        ctx.name = name
        ctx.bwd = bwd
        with torch.no_grad():
            out = fn(*tensors)
        outs = (out,) if not isinstance(out, tuple) else out
        ctx.save_for_backward(*tensors)
        ctx.out_meta = tuple(None if o is None else OutputMeta(o) for o in outs)
        nd = [outs[i] for i in nondiff if i < len(outs) and outs[i] is not None]
        nd += [o for o in outs if o is not None and not o.dtype.is_floating_point]
        if nd:
            ctx.mark_non_differentiable(*nd)
        return out

    @staticmethod
    def backward(ctx, *grads):
        if torch.is_grad_enabled():
            # the engine runs backward functions with grad mode ON only under create_graph=True.  The HIP backward
            # kernels record no graph, so the node would silently act as a constant in the second-order graph
            # (once_differentiable does not catch it either when the incoming gradients carry no graph themselves)
            raise RuntimeError(
                'nerf_from_image_amd: %s is first-order only - torch.autograd.grad(..., create_graph=True) / a double '
                'backward through a HIP node is not supported (the reference needs it for the path-length output of '
                'the plane producer only; keep HIP nodes out of that graph, see handoff.unfused)' % ctx.name)
        if ctx.bwd is None:
            raise NotImplementedError(
                'nerf_from_image_amd: %s has no HIP backward (forward-only op); wrap the call in '
                'torch.no_grad() or detach its inputs' % ctx.name)
        with torch.no_grad():
            gin = ctx.bwd(ctx.saved_tensors, ctx.out_meta, grads, ctx.needs_input_grad[4:])
        gin = tuple(g if need else None for g, need in zip(gin, ctx.needs_input_grad[4:]))
        return (None, None, None, None) + gin


def differentiable(name, fn, *tensors, bwd=None, non_differentiable_outputs=()):
    """Runs fn(*tensors) (HIP kernels).  Records an autograd node only when a gradient can be asked for.

    bwd(inputs, output_meta, grad_outputs, needs) -> tuple of gradients, one per input (None allowed);
    output_meta[i] is an :class:`OutputMeta` (shape / dtype / device of output i), never the tensor."""
    needs = torch.is_grad_enabled() and any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors)
    if not needs:
        with torch.no_grad():
            return fn(*tensors)
    return _HipNode.apply(name, fn, bwd, tuple(non_differentiable_outputs), *tensors)


def zeros_like_or(g, ref):
    """Autograd hands None for outputs nobody used; kernels want dense tensors.  ref: tensor or OutputMeta."""
    if g is None:
        return ref.zeros() if isinstance(ref, OutputMeta) else torch.zeros_like(ref)
    return g.contiguous()
