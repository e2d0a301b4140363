"""Lifetime of the tensors an autograd node of nerf_from_image_amd.autograd keeps: outputs must
not be reachable from the node (cycle output -> grad_fn -> ctx -> output), inputs are saved the autograd way."""
import gc
import weakref

import pytest
import torch

from nerf_from_image_amd.autograd import OutputMeta, differentiable, zeros_like_or


def _op(x, w):
    def fwd(a, b):
        return a * b, (a + b).sum(dim=-1)

    def bwd(inputs, out_meta, grads, needs):
        a, b = inputs
        assert isinstance(out_meta[0], OutputMeta) and out_meta[0].shape == tuple(a.shape)
        g0 = zeros_like_or(grads[0], out_meta[0])
        g1 = zeros_like_or(grads[1], out_meta[1])
        return g0 * b + g1[..., None], g0 * a + g1[..., None]
    return differentiable('test_op', fwd, x, w, bwd=bwd)


def test_outputs_die_after_backward():
    x = torch.randn(5, 3, requires_grad=True)
    w = torch.randn(5, 3, requires_grad=True)
    y, z = _op(x, w)
    ry, rz = weakref.ref(y), weakref.ref(z)
    (y.sum() + 2 * z.sum()).backward()
    assert torch.allclose(x.grad, w.detach() + 2) and torch.allclose(w.grad, x.detach() + 2)
    del y, z
    gc.collect()
    assert ry() is None and rz() is None


def test_outputs_die_without_backward():
    x = torch.randn(4, 2, requires_grad=True)
    y, z = _op(x, torch.randn(4, 2))
    ry = weakref.ref(y)
    del y, z
    gc.collect()
    assert ry() is None


def test_retain_graph_allows_two_backwards():
    x = torch.randn(3, 2, requires_grad=True)
    w = torch.randn(3, 2)
    y, _ = _op(x, w)
    y.sum().backward(retain_graph=True)
    y.sum().backward()
    assert torch.allclose(x.grad, 2 * w)


def test_forward_only_op_raises_on_backward():
    x = torch.randn(3, requires_grad=True)
    y = differentiable('fwd_only', lambda a: a * 2, x)
    with pytest.raises(NotImplementedError):
        y.sum().backward()


@pytest.mark.gpu
def test_device_memory_is_flat_over_steps(gpu_device):
    """A differentiable render step must not grow the allocator's live set from step to step."""
    import types
    from stand_in import StandInGenerator, look_at_cameras
    import nerf_from_image_amd.generator as nfi_gen
    import nerf_from_image_amd.render as nfi_render
    torch.manual_seed(0)
    model = nfi_gen.attach(StandInGenerator(0.55, attention_values=10, use_sdf=True, plane_res=48).to(gpu_device).train())
    g = torch.Generator().manual_seed(1)
    cam = look_at_cameras(2, 1.6, g).to(gpu_device)
    focal = torch.full((2,), 1.0254, device=gpu_device)
    z = torch.randn(2, 512, generator=g).to(gpu_device)
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=True)
    render = nfi_render.make_render(cfg, {'scene_range': 0.55, 'white_background': True})
    params = [p for p in model.parameters() if p.requires_grad]
    live = []
    for step in range(6):
        rgb, _, mask, _, _, _ = render(model, 32, 32, cam, focal, None, None, z, 32)
        (rgb.mean() + mask.mean()).backward()
        for p in params:
            p.grad = None
        del rgb, mask
        torch.cuda.synchronize()
        live.append(torch.cuda.memory_allocated(gpu_device))
    assert live[-1] == live[2], live


def test_double_backward_through_a_hip_node_raises():
    """The HIP nodes are first-order only: create_graph=True through one must fail loudly, not return a constant."""
    import pytest
    from nerf_from_image_amd.autograd import differentiable
    x = torch.randn(4, requires_grad=True)
    y = differentiable('square', lambda t: t * t, x, bwd=lambda inputs, meta, grads, needs: (2 * inputs[0] * grads[0],))
    with pytest.raises(RuntimeError, match='first-order only'):
        torch.autograd.grad(y.sum(), x, create_graph=True)
    g, = torch.autograd.grad(y.sum(), x)
    assert torch.allclose(g, 2 * x.detach()) and not g.requires_grad


def test_path_length_request_runs_the_unfused_block_tail():
    """generator.hip_forward / wrapped_forward keep the fused hand-off node (first-order only) out of a forward that
    asks for the second-order 'path_length' output."""
    from nerf_from_image_amd import handoff

    class Block(torch.nn.Module):
        conv1 = torgb = None
        in_channels = 0

        def forward(self, *a):
            return 'original'

    class Net(torch.nn.Module):
        img_resolution = 8

        def __init__(self):
            super().__init__()
            self.b8 = Block()
    net = Net()
    blk = handoff.last_block(net)
    blk._nfi_original_forward = blk.forward
    blk.forward = lambda *a: 'fused'
    assert net.b8.forward() == 'fused'
    with handoff.unfused(net):
        assert net.b8.forward() == 'original'
    assert net.b8.forward() == 'fused'
    with handoff.unfused(torch.nn.Linear(2, 2)):        # not a StyleGAN2-style network: no-op
        pass
