import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
from nerf_from_image_amd import field_backward as fb, ops
from oracle import nfi_oracle as orc
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(70500)
B, A, r, P, res = 2, 10, 0.55, 70000, 64
planes = torch.randn(B, 3, 32, res, res, generator=g).to(dev)
w1 = torch.randn(64, 32, generator=g).to(dev); b1 = (0.3*torch.randn(64, generator=g)).to(dev)
w2 = torch.randn(11, 64, generator=g).to(dev); b2 = (0.3*torch.randn(11, generator=g)).to(dev)
x = ((torch.rand(B, P, 3, generator=g) * 2 - 1) * r * 1.15).to(dev)
att = (torch.rand(B, A, 3, generator=g) * 2 - 1).to(dev)
beta, alpha = torch.tensor([0.12], device=dev), torch.tensor([0.3], device=dev)
gs = torch.randn(B, P, generator=g).to(dev); gr = torch.randn(B, P, 3, generator=g).to(dev)
texels = ops.planes_to_texels(planes); image = ops.decoder_pack(w1, b1, w2, b2, A)
runs = {}
for name, mode in (('a0', 0), ('b0', 0), ('a1', 1), ('b1', 1)):
    runs[name] = fb.field_query_bwd(x, texels, image, w1, w2, r, A, att, True, beta, alpha, gs, gr, want_points=True, scatter_mode=mode)['g_points'].clone()
for a, b in (('a0', 'b0'), ('a1', 'b1'), ('a0', 'a1')):
    d = (runs[a] - runs[b]).abs()
    print(a, b, 'max diff', d.max().item(), 'n diff', int((d > 0).sum()), 'of', d.numel(), 'argmax', int(d.argmax()))
# oracle in float64
xc = x.cpu().double().requires_grad_()
q = orc.field_query(planes.cpu().double(), w1.cpu().double(), b1.cpu().double(), w2.cpu().double(), b2.cpu().double(), xc, r, True,
                    beta.cpu().double(), alpha.cpu().double(), att.cpu().double())
ref, = torch.autograd.grad((q['sigma'] * gs.cpu().double()).sum() + (q['rgb'] * gr.cpu().double()).sum(), xc)
for n in ('a0', 'a1'):
    d = (runs[n].cpu().double() - ref).abs()
    print(n, 'vs oracle: max', d.max().item(), 'scale', ref.abs().max().item(), 'n > 1e-3*scale', int((d > 1e-3 * ref.abs().max()).sum()))
    idx = d.flatten().topk(5).indices
    for i in idx:
        p = int(i) // 3
        print('   point', p % P, 'scene', p // P, 'lane', (p % P) % 64, 'tile', ((p % P) % 64) // 16, 'hip', runs[n].flatten()[i].item(), 'ref', ref.flatten()[i].item())
