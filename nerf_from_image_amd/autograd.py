"""Autograd glue between the HIP kernels and PyTorch.

Every host-level op goes through :func:`differentiable`: the forward runs the HIP kernel(s); if
any input requires grad, the call is recorded as one autograd node whose backward is looked up in
``BACKWARD`` (filled in by the modules that own backward kernels).  An op without a registered
backward fails loudly when a gradient is actually requested, it never falls back to ATen.
"""
import torch

BACKWARD = {}


def register_backward(name):
    def deco(fn):
        BACKWARD[name] = fn
        return fn
    return deco


class _HipNode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, name, fn, nondiff, n_in, *tensors):
        ctx.name = name
        ctx.present = [t is not None for t in tensors]
        with torch.no_grad():
            out = fn(*tensors)
        single = not isinstance(out, tuple)
        outs = (out,) if single else out
        ctx.single = single
        ctx.saved_inputs = tensors
        ctx.saved_outputs = outs
        nd = [outs[i] for i in nondiff if i < len(outs) and outs[i] is not None]
        nd += [o for o in outs if o is not None and not o.dtype.is_floating_point]
        if nd:
            ctx.mark_non_differentiable(*nd)
        return out

    @staticmethod
    def backward(ctx, *grads):
        bw = BACKWARD.get(ctx.name)
        if bw is None:
            raise NotImplementedError(
                'nerf_from_image_amd: no HIP backward is registered for %s yet (forward-only op); '
                'wrap the call in torch.no_grad() or detach its inputs' % ctx.name)
        gin = bw(ctx, *grads)
        return (None, None, None, None) + tuple(gin)


def differentiable(name, fn, *tensors, non_differentiable_outputs=()):
    """Runs fn(*tensors) (HIP kernels).  Records an autograd node only when needed."""
    needs = torch.is_grad_enabled() and any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors)
    if not needs:
        with torch.no_grad():
            return fn(*tensors)
    return _HipNode.apply(name, fn, tuple(non_differentiable_outputs), len(tensors), *tensors)
