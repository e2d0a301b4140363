"""ISA-level variants of field_query_bwd_kernel<true,true,false,0> for the packed-fp32 wrong-product hunt: the SLP build
of nfi_backward_field.hip is compiled to device assembly ONCE, single instructions of the failing block are padded /
replaced in the TEXT, and every variant is assembled and linked into build/variants/libnfi_isa_<name>.so - register
allocation and scheduling of everything else stay byte for byte what the compiler produced.

    python tools/probes/isa_patch_variants.py            # here (no GPU)
    python tools/probes/bwd_variants.py run 300 isa_base isa_nop_before_mul ...      # on the GPU box
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, 'nerf_from_image_amd', 'csrc')
OUT = os.path.join(ROOT, 'build', 'variants')
LLVM = '/opt/rocm/lib/llvm/bin'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC']
KERNEL = '_Z22field_query_bwd_kernelILb1ELb1ELb0ELi0EEv14FieldBwdParams'
MUL = '\tv_pk_mul_f32 v[10:11], v[10:11], v[126:127] op_sel:[0,1] op_sel_hi:[1,0]\n'
MUL_PREV = '\tv_pk_mul_f32 v[54:55], v[128:129], v[54:55] op_sel:[1,0] op_sel_hi:[0,1]\n'
DIFF = '\tv_pk_add_f32 v[10:11], v[10:11], v[58:59] neg_lo:[0,1] neg_hi:[0,1]\n'
SUM = '\tv_pk_add_f32 v[10:11], v[10:11], v[54:55]\n'

PATCHES = {
    'isa_base': lambda k: k,
    'isa_nop_before_mul': lambda k: k.replace(MUL, '\ts_nop 3\n' + MUL),
    'isa_nop1_before_mul': lambda k: k.replace(MUL, '\ts_nop 0\n' + MUL),
    'isa_nop_after_mul': lambda k: k.replace(MUL, MUL + '\ts_nop 3\n'),
    'isa_nop_before_diff': lambda k: k.replace(DIFF, '\ts_nop 3\n' + DIFF),
    'isa_scalar_mul': lambda k: k.replace(MUL, '\tv_mul_f32_e32 v10, v10, v127\n\tv_mul_f32_e32 v11, v11, v126\n'),
    'isa_swap_muls': lambda k: k.replace(MUL_PREV + MUL, MUL + MUL_PREV),
    'isa_waitcnt_before_mul': lambda k: k.replace(MUL, '\ts_waitcnt vmcnt(0) lgkmcnt(0)\n' + MUL),
    'isa_nop_before_sum': lambda k: k.replace(SUM, '\ts_nop 3\n' + SUM),
}


def sh(*cmd):
    subprocess.check_call(list(cmd))


def main():
    os.makedirs(OUT, exist_ok=True)
    work = os.path.join(OUT, 'isa')
    os.makedirs(work, exist_ok=True)
    src = os.path.join(CSRC, 'nfi_backward_field.hip')
    dev_s = os.path.join(work, 'dev.s')
    sh('/opt/rocm/bin/hipcc', *FLAGS, '-S', '--cuda-device-only', src, '-o', dev_s)
    text = open(dev_s).read()
    a = text.index(KERNEL + ':')
    b = text.index('s_endpgm', a)
    kern = text[a:b]
    for pat in (MUL, MUL_PREV + MUL, DIFF, SUM):
        assert kern.count(pat) == 1, 'block not found exactly once: %r (%d)' % (pat, kern.count(pat))
    for name, fn in PATCHES.items():
        new = fn(kern)
        assert name == 'isa_base' or new != kern, name
        s_path = os.path.join(work, name + '.s')
        open(s_path, 'w').write(text[:a] + new + text[b:])
        o, out, fb, host = (os.path.join(work, name + e) for e in ('.dev.o', '.out', '.hipfb', '.o'))
        sh(LLVM + '/clang', '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', s_path, '-o', o)
        sh(LLVM + '/lld', '-flavor', 'gnu', '-m', 'elf64_amdgpu', '--no-undefined', '-shared', '-o', out, o)
        sh(LLVM + '/clang-offload-bundler', '-type=o', '-bundle-align=4096',
           '-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950', '-input=/dev/null', '-input=' + out,
           '-output=' + fb)
        sh('/opt/rocm/bin/hipcc', *FLAGS, '--cuda-host-only', '-Xclang', '-fcuda-include-gpubinary', '-Xclang', fb, '-c', src,
           '-o', host)
        sh('/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-fPIC', '-shared', os.path.join(ROOT, 'build', 'nfi_kernels.o'), host,
           '-o', os.path.join(OUT, 'libnfi_%s.so' % name))
        print('built', name, flush=True)


if __name__ == '__main__':
    main()
