"""Determinism probe of field_query_bwd's g_points: for every corrupted point, is the wrong value another point's
correct value (stale LDS row / stale register) or garbage?"""
import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from nerf_from_image_amd import field_backward as fb, ops
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
modes = sys.argv[2] if len(sys.argv) > 2 else '01p'
def run(mode, **kw):
    return fb.field_query_bwd(x, texels, image, w1, w2, r, A, att, True, beta, alpha, gs, gr, scatter_mode=mode, **kw)
# majority reference over 5 runs
runs = torch.stack([run(0, want_points=True)['g_points'].clone() for _ in range(5)])
ref = runs.median(dim=0).values
print('reference runs disagreeing with the median:', [int(((q - ref).abs().amax(-1) > 0).sum()) for q in runs])
ev = 0
for it in range(N):
    for m in modes:
        if m == 'p':
            continue
        got = run(int(m), want_points=True)['g_points']
        d = (got - ref).abs().amax(dim=-1)
        for s, p in torch.nonzero(d > 0).tolist():
            ev += 1
            gv, rv = got[s, p], ref[s, p]
            # whose value is it?
            lo, hi = max(0, p - 4096), min(P, p + 4096)
            near = ref[s, lo:hi]
            match = torch.nonzero((near == gv).all(-1)).flatten().tolist()
            comp = [(float(gv[c]), float(rv[c])) for c in range(3)]
            print('it %2d mode %s scene %d point %6d chunk %5d tile %d pt %2d  got/ref %s  equals ref of points %s' % (
                it, m, s, p, p // 64, (p % 64) // 16, p % 16, ' '.join('%.5g/%.5g' % c for c in comp), [lo + i for i in match][:4]))
print('events', ev, 'in', N, 'iterations of modes', modes)
