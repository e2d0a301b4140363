"""Per-kernel register / scratch / LDS / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage`.
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c nerf_from_image_amd/csrc/nfi_kernels.hip \
        -o /tmp/nfi.o -Rpass-analysis=kernel-resource-usage 2> /tmp/res_usage.txt
  python tools/resource_usage.py /tmp/res_usage.txt [substring ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
keys = sys.argv[2:]
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
names = [b.split('\n')[0].strip() for b in blocks]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
print('%-96s %4s %4s %5s %3s %4s %6s' % ('kernel', 'vgpr', 'agpr', 'scr B', 'occ', 'sgpr', 'lds B'))
for b, nm in zip(blocks, dem):
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    nm = re.sub(r'^void ', '', nm)
    nm = re.sub(r'\(.*$', '', nm)
    if keys and not any(k in nm for k in keys):
        continue
    print('%-96s %4d %4d %5d %3d %4d %6d' % (nm[:96], g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'),
                                            g(r'Occupancy \[waves/SIMD\]'), g('SGPRs'), g(r'LDS Size \[bytes/block\]')))
