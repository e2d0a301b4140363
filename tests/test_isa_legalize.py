"""CPU tests of the build step that removes the packed-fp32 operand form MI355X computes wrongly next to a K=32 16-bit
MFMA (tools/gfx950_pk_legalize.py; HISTORY.md "Determinism"; probe: tools/probes/pk_hazard.hip)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import gfx950_pk_legalize as legal  # noqa: E402


def pk_eval(line, regs):
    """Tiny interpreter of v_pk_{mul,add,fma}_f32 on symbolic register pairs: returns (dst, (lo expr, hi expr))."""
    indent, op, gap, operands, mods, tail, n = legal._parse(line)
    sel = mods.get('op_sel', [0] * n)
    hi = mods.get('op_sel_hi', [1] * n)
    nlo = mods.get('neg_lo', [0] * n)
    nhi = mods.get('neg_hi', [0] * n)

    def src(i, half_sel, neg):
        v = regs[operands[1 + i]][half_sel]
        return ('-' + v) if neg else v
    lo = [src(i, sel[i], nlo[i]) for i in range(n)]
    hh = [src(i, hi[i], nhi[i]) for i in range(n)]

    def combine(x):
        core = frozenset(x[:2])                      # add / mul / the product of fma commute
        return (op, core, x[2] if n == 3 else None)
    return operands[0], (combine(lo), combine(hh))


def test_rewrite_preserves_semantics_and_clears_the_bad_form():
    regs = {'v[10:11]': ('a0', 'a1'), 'v[126:127]': ('b0', 'b1'), 'v[2:3]': ('c0', 'c1'), 's[4:5]': ('s0', 's1')}
    bad_lines = [
        '\tv_pk_mul_f32 v[10:11], v[10:11], v[126:127] op_sel:[0,1] op_sel_hi:[1,0]',
        '\tv_pk_mul_f32 v[2:3], v[10:11], v[126:127] op_sel:[0,1]',
        '\tv_pk_add_f32 v[2:3], v[10:11], v[126:127] op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[0,1]',
        '\tv_pk_fma_f32 v[2:3], v[10:11], v[126:127], v[2:3] op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]',
        '\tv_pk_add_f32 v[2:3], s[4:5], v[126:127] op_sel:[0,1] op_sel_hi:[1,0] ; a comment',
    ]
    for line in bad_lines:
        assert legal.audit(line)[0], line
        new, n = legal.legalize(line)
        assert n == 1 and not legal.audit(new)[0], new
        assert pk_eval(line, regs) == pk_eval(new, regs), (line, new)
        # the mirrored form reads src0.high and src1.low for the low result
        p = legal._parse(new)
        assert p[4]['op_sel'][:2] == [1, 0]


def test_forms_measured_correct_are_left_alone():
    ok = [
        '\tv_pk_mul_f32 v[2:3], v[10:11], v[126:127]',
        '\tv_pk_mul_f32 v[2:3], v[10:11], v[126:127] op_sel:[1,0] op_sel_hi:[0,1]',
        '\tv_pk_mul_f32 v[2:3], v[10:11], v[126:127] op_sel:[1,1] op_sel_hi:[0,0]',
        '\tv_pk_mul_f32 v[2:3], v[10:11], v[126:127] op_sel_hi:[0,1]',
        '\tv_pk_mul_f32 v[2:3], v[10:11], v[126:127] op_sel_hi:[1,0]',
        '\tv_pk_add_f32 v[12:13], v[10:11], v[10:11] op_sel:[0,1] op_sel_hi:[1,0]',      # src0 == src1
        '\tv_pk_add_f32 v[2:3], v[10:11], v[126:127] op_sel:[1,1] op_sel_hi:[1,0]',
        '\tv_pk_fma_f32 v[2:3], v[10:11], v[126:127], v[2:3] op_sel:[0,0,1] op_sel_hi:[1,1,0]',
        '\tv_pk_fma_f32 v[2:3], v[10:11], v[126:127], v[2:3] op_sel:[1,0,0]',
        '\tv_pk_mov_b32 v[2:3], v[10:11], v[126:127] op_sel:[0,1]',
        '\tv_pk_add_f32 v[128:129], v[126:127], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]',
    ]
    text = '\n'.join(ok)
    new, n = legal.legalize(text)
    assert n == 0 and new == text
    assert not legal.audit(text)[0]


def test_audit_names_the_kernel():
    text = 'my_kernel:\n\ts_nop 0\n\tv_pk_mul_f32 v[2:3], v[10:11], v[126:127] op_sel:[0,1]\n.LBB0_1:\n\ts_endpgm\n'
    bad, seen = legal.audit(text)
    assert seen == 1 and bad == [('my_kernel', 3, 'v_pk_mul_f32 v[2:3], v[10:11], v[126:127] op_sel:[0,1]')]


def test_the_built_library_went_through_the_legaliser():
    """build() writes build/isa_audit.txt: one line per translation unit with the instructions seen / rewritten."""
    report = os.path.join(ROOT, 'build', 'isa_audit.txt')
    if not os.path.exists(report):
        import pytest
        pytest.skip('library not built here')
    lines = open(report).read().strip().splitlines()
    assert lines[0].startswith('compiler:')
    assert len(lines) == 3 and all('0 of the bad form left' in ln for ln in lines[1:])
