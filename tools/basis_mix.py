"""out[b] = sum_k coef[b, k] * basis[k]: the stand-in plane producers' only heavy operation (tests/stand_in.py,
tools/train_bench.py), with a backward that stays memory bound.

torch.einsum('bk,kchw->bchw') hands the coefficient gradient - a [B x N] x [N x K] product with B = K = 4 and
N = 6.3 M - to a rocBLAS GEMM kernel that takes 1.84 ms on MI355X (profiles/r2, training step), a fifth of the step
the stand-in is supposed to stay out of the way of.  K matrix-vector products read the same 200 MB in 0.15 ms."""
import torch


class _BasisMix(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coef, basis):
        ctx.save_for_backward(coef, basis)
        k = basis.shape[0]
        return (coef @ basis.reshape(k, -1)).view(coef.shape[0], *basis.shape[1:])

    @staticmethod
    def backward(ctx, g):
        coef, basis = ctx.saved_tensors
        k = basis.shape[0]
        g2 = g.reshape(g.shape[0], -1)
        g_coef = g_basis = None
        if ctx.needs_input_grad[0]:
            flat = basis.reshape(k, -1)
            g_coef = torch.stack([torch.mv(g2, flat[i]) for i in range(k)], dim=1)
        if ctx.needs_input_grad[1]:
            g_basis = (coef.t() @ g2).view_as(basis)
        return g_coef, g_basis


def basis_mix(coef, basis):
    """coef [B, K], basis [K, ...] -> [B, ...]"""
    return _BasisMix.apply(coef, basis)
