"""CPU test of the Generator wrapper against the LIVE reference class (needs the reference sources: the
checkout, or the copy oracle/make_ref.py staged): the original forward keeps producing path-length / regulariser outputs, the planes and the
attention table captured for the HIP sampler are exactly what the reference's own sampler would use."""
import os
import sys

import pytest
import torch

from oracle import reference

REF = reference.root()          # the checkout, or the copy oracle/make_ref.py staged
pytestmark = pytest.mark.skipif(REF is None, reason='reference sources not available (oracle/make_ref.py)')


def test_wrapped_forward_keeps_reference_outputs_and_captures_planes(monkeypatch):
    ref_gen = reference.modules().generator
    import nerf_from_image_amd.generator as nfi_gen
    torch.manual_seed(0)
    model = ref_gen.Generator(512, 0.55, attention_values=10, use_sdf=True, disable_stylegan_noise=True)
    keys_before = list(model.state_dict().keys())
    seen = {}

    def fake_make_sampler(planes, decoder, scene_range, n_att, att, use_sdf, beta, alpha, texel_dtype=0,
                          request_model_outputs=(), viewdir=None, texel_cache=None):
        assert viewdir is None
        seen.update(planes=planes, att=att, decoder=decoder, beta=beta, alpha=alpha, scene_range=scene_range)
        return lambda x, req=['sigma', 'rgb']: {'sigma': None}
    monkeypatch.setattr(nfi_gen, 'make_sampler', fake_make_sampler)
    nfi_gen.attach(model)
    assert list(model.state_dict().keys()) == keys_before          # checkpoints stay loadable
    z = torch.randn(2, 512)
    out = model(None, z, ['sampler', 'path_length', 'sdf_eikonal_loss', 'sdf_distance_loss'])
    assert set(out) == {'sampler', 'path_length', 'sdf_eikonal_loss', 'sdf_distance_loss'}
    assert out['path_length'].shape == (2,) and out['sdf_eikonal_loss'].requires_grad
    assert seen['planes'].shape == (2, 3, 32, 256, 256) and seen['att'].shape == (2, 10, 3)
    assert seen['decoder'] is model.decoder and seen['beta'] is model.beta and seen['scene_range'] == 0.55
    # the captured planes / attention values are the ones the reference's own sampler closes over
    torch.manual_seed(1)
    model2 = ref_gen.Generator(512, 0.55, attention_values=10, use_sdf=True, disable_stylegan_noise=True)
    model2.load_state_dict(model.state_dict())
    model2.eval(); model.eval()
    with torch.no_grad():
        ref_out = model2(None, z, ['sampler', 'attention_values'])
        model(None, z, ['sampler'])
        x = torch.rand(2, 50, 3) - 0.5
        ref_sig = ref_out['sampler'](x, ['sigma', 'rgb'])
        from oracle import nfi_oracle as orc
        dec = model.decoder.net
        mine = orc.field_query(seen['planes'], dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias, x, 0.55, True,
                               model.beta, model.alpha, seen['att'])
    assert torch.allclose(mine['sigma'], ref_sig['sigma'], atol=1e-6)
    assert torch.allclose(mine['rgb'], ref_sig['rgb'], atol=1e-6)
    assert torch.equal(seen['att'], ref_out['attention_values'])
    # attention override passes through the original forward
    att_over = torch.rand(2, 10, 3)
    with torch.no_grad():
        model(None, z, ['sampler'], {'attention_values': att_over})
    assert torch.equal(seen['att'], att_over)


def test_wrapped_forward_hands_over_the_view_direction_feature(monkeypatch):
    """--use_viewdir: the per-ray feature captured from ViewDirectionMapper.fc6 and the mapper's output layer,
    fed to the oracle's restatement of the closure, reproduce the reference's own sampler."""
    ref_gen = reference.modules().generator
    import nerf_from_image_amd.generator as nfi_gen
    from oracle import nfi_oracle as orc
    torch.manual_seed(0)
    model = ref_gen.Generator(512, 0.55, attention_values=10, use_sdf=True, disable_stylegan_noise=True, use_viewdir=True)
    with torch.no_grad():
        model.viewdir_mapper.output.weight.normal_()
        model.viewdir_mapper.output.bias.normal_()
    model.eval()
    seen = {}

    def fake_make_sampler(planes, decoder, scene_range, n_att, att, use_sdf, beta, alpha, texel_dtype=0,
                          request_model_outputs=(), viewdir=None, texel_cache=None):
        seen.update(planes=planes, att=att, viewdir=viewdir)
        return lambda x, req=['sigma', 'rgb']: {}
    monkeypatch.setattr(nfi_gen, 'make_sampler', fake_make_sampler)
    nfi_gen.attach(model)
    z = torch.randn(2, 512)
    H = W = 3
    S = 5
    viewdirs = torch.nn.functional.normalize(torch.randn(2, H, W, 1, 3), dim=-1)
    x = torch.rand(2, H, W, S, 3) - 0.5
    with torch.no_grad():
        model(viewdirs, z, ['sampler'])
        ref = model._nfi_original_forward(viewdirs, z, ['sampler'])['sampler'](x, ['sigma', 'rgb'])
        ray_feature, out_layer = seen['viewdir']
        assert ray_feature.shape == (2, H, W, 1, 32) and out_layer is model.viewdir_mapper.output
        dec = model.decoder.net
        mine = orc.field_query(seen['planes'], dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias, x, 0.55, True,
                               model.beta, model.alpha, seen['att'],
                               viewdir=dict(x=ray_feature.reshape(2, H * W, 32), w3=out_layer.weight, b3=out_layer.bias))
    assert torch.allclose(mine['sigma'], ref['sigma'], atol=1e-6)
    assert torch.allclose(mine['rgb'], ref['rgb'], atol=1e-6)


def test_oracle_bbox_overlay_matches_reference():
    """The 'bbox' visualisation overlay (generator.py:645-659) restated in the oracle, against the live closure."""
    ref_gen = reference.modules().generator
    from oracle import nfi_oracle as orc
    torch.manual_seed(3)
    model = ref_gen.Generator(512, 0.55, attention_values=10, use_sdf=True, disable_stylegan_noise=True).eval()
    z = torch.randn(1, 512)
    x = (torch.rand(1, 6, 6, 40, 3) * 2 - 1) * 0.6          # inside, near the faces / edges, and outside
    with torch.no_grad():
        out = model(None, z, ['sampler', 'bbox'])
        ref = out['sampler'](x, ['sigma', 'coords'])
        plain = model(None, z, ['sampler'])['sampler'](x, ['sigma'])['sigma']
        outside = ((x.view(1, -1, 3) / 0.55).abs() > 1).any(dim=-1).float()
        mine = orc.bbox_overlay(x, plain, outside, 0.55)
    assert torch.equal(ref['coords'], x)
    assert torch.equal(mine, ref['sigma'])
    assert (mine != plain).float().mean() > 0.01


def test_oracle_regularisers_match_reference():
    """The regulariser branch (generator.py:505-585: eikonal with its double backward, distance, total variation,
    entropy) restated in the oracle against the live Generator: losses AND their gradients w.r.t. the decoder."""
    ref_gen = reference.modules().generator
    from oracle import nfi_oracle as orc
    torch.manual_seed(5)
    model = ref_gen.Generator(512, 0.55, attention_values=10, use_sdf=True, disable_stylegan_noise=True).train()
    z = torch.randn(2, 512)
    cap, draws = {}, {}
    hook = model.synthesis_network.register_forward_hook(lambda m, i, o: cap.__setitem__('planes', o))
    real_rand_like, real_randn_like = torch.rand_like, torch.randn_like

    def rand_like(t, **k):
        draws['jitter'] = real_rand_like(t, **k)
        return draws['jitter']

    def randn_like(t, **k):
        draws['perturb'] = real_randn_like(t, **k)
        return draws['perturb']
    torch.rand_like, torch.randn_like = rand_like, randn_like
    try:
        ref = model(None, z, ['sdf_eikonal_loss', 'sdf_distance_loss', 'total_variation_loss', 'entropy_loss'])
    finally:
        torch.rand_like, torch.randn_like = real_rand_like, real_randn_like
        hook.remove()
    names = ['sdf_eikonal_loss', 'sdf_distance_loss', 'total_variation_loss', 'entropy_loss']
    dec = model.decoder.net
    params = [dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias]
    ref_g = torch.autograd.grad(sum(ref[n].sum() for n in names), params, retain_graph=True)

    planes = cap['planes'].detach().view(2, 3, 32, 256, 256)
    bins = orc.stratified_volume(2, 32, 0.55, draws['jitter'])
    mine = orc.regularisers(planes, *params, bins, 0.55, True, model.beta, draws['perturb'])
    for n in names:
        assert torch.allclose(mine[n], ref[n], rtol=1e-5, atol=1e-7), (n, mine[n], ref[n])
    my_g = torch.autograd.grad(sum(mine[n].sum() for n in names), params)
    for a, b in zip(my_g, ref_g):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-7 * float(b.abs().max() + 1)), (a - b).abs().max()


def test_wrapped_forward_can_hand_the_regularisers_to_hip(monkeypatch):
    """attach(..., hip_regularisers=True): the four regulariser names are withheld from the reference forward (which
    still produces the planes) and served by generator.regulariser_outputs on the captured planes."""
    ref_gen = reference.modules().generator
    import nerf_from_image_amd.generator as nfi_gen
    torch.manual_seed(0)
    model = ref_gen.Generator(512, 0.55, attention_values=10, use_sdf=True, disable_stylegan_noise=True).train()
    seen = {}

    def fake_regularisers(self, planes, request, texel_cache=None):
        seen.update(planes=planes, request=list(request))
        return {r: torch.zeros(planes.shape[0]) for r in request if r.endswith('_loss')}
    monkeypatch.setattr(nfi_gen, 'regulariser_outputs', fake_regularisers)
    nfi_gen.attach(model, hip_regularisers=True)
    inner = {}
    orig = model._nfi_original_forward
    model._nfi_original_forward = lambda v, c, req, mi: (inner.__setitem__('req', list(req)) or orig(v, c, req, mi))
    monkeypatch.setattr(nfi_gen, 'make_sampler', lambda *a, **k: (lambda x, req=['sigma', 'rgb']: {}))
    out = model(None, torch.randn(2, 512), ['sampler', 'sdf_distance_loss', 'sdf_eikonal_loss', 'path_length'])
    assert inner['req'] == ['sampler', 'path_length', 'attention_values']
    assert set(out) == {'sampler', 'sdf_distance_loss', 'sdf_eikonal_loss', 'path_length'}
    assert seen['planes'].shape == (2, 3, 32, 256, 256) and seen['planes'].requires_grad


def test_a_deep_copy_of_an_attached_model_is_attached_to_itself(monkeypatch):
    """run.py keeps EMA / test copies of its generator (copy.deepcopy, load_state_dict into a second instance): the copy's
    forward, its kept original forward and the fused last block have to be bound to the COPY's modules and parameters."""
    import copy
    ref_gen = reference.modules().generator
    import nerf_from_image_amd.generator as nfi_gen
    torch.manual_seed(0)
    model = nfi_gen.attach(ref_gen.Generator(512, 0.55, attention_values=10, use_sdf=True, disable_stylegan_noise=True),
                           hip_regularisers=True, fused_handoff=True)
    twin = copy.deepcopy(model)
    assert twin.forward.__self__ is twin and twin._nfi_original_forward.__self__ is twin
    assert model.forward.__self__ is model
    last = [b for b in twin.synthesis_network.children() if hasattr(b, '_nfi_original_forward')]
    assert len(last) == 1 and last[0].forward.__self__ is last[0] and last[0]._nfi_original_forward.__self__ is last[0]
    assert twin.nfi_hip_regularisers and list(twin.state_dict()) == list(model.state_dict())
    seen = []
    monkeypatch.setattr(nfi_gen, 'make_sampler', lambda planes, decoder, *a, **k: seen.append(decoder) or (lambda x, req=None: {}))
    from nerf_from_image_amd import handoff
    with torch.no_grad(), handoff.unfused(twin.synthesis_network):           # (the fused tail is a HIP kernel; the CPU runs the original)
        twin.eval()(None, torch.randn(1, 512), ['sampler'])
    assert seen == [twin.decoder] and seen[0] is not model.decoder
