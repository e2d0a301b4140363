"""Does the Infinity Cache (256 MB, memory side) serve a re-read of a buffer that left the L2s faster than HBM does?
Reads (torch.sum) of ONE buffer of N MB over and over against a cycle over buffers totalling 1 GB (which cannot stay in
the cache), for N = 32 ... 512.  The question behind it: would the plane-gradient scatter (bin_reduce_kernel: every
128-B gradient row read once per plane, 3.6 GB per training step at 80 % of the HBM rate) gain from an ordering that
brings a row's three reads within one cache-resident window?  GPU box: python tools/probes/mall_probe.py"""
import torch

dev = torch.device('cuda:0')
for mb in (32, 64, 128, 192, 256, 512):
    n = mb * (1 << 20) // 4
    same = [torch.randn(n, device=dev)]
    many = [torch.randn(n, device=dev) for _ in range(max(2, 1024 // mb))]
    for name, bufs in (('same buffer', same), ('cycle over %d buffers' % len(many), many)):
        reps = max(8, 4096 // mb)
        for i in range(4):
            bufs[i % len(bufs)].sum()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(reps):
            bufs[i % len(bufs)].sum()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        print('%4d MB  %-24s %.4f ms per read  %.2f TB/s' % (mb, name, ms, mb * (1 << 20) / ms / 1e9), flush=True)
