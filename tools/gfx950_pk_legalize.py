"""Build step: remove the packed-fp32 operand form that gfx950 (MI355X) computes wrongly next to a K=32 16-bit MFMA.

Finding (tools/probes/pk_hazard.hip, profiles/r3/pk_hazard.log; HISTORY.md "Determinism"): a VOP3P packed-fp32 instruction
(v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) whose LOW result half reads src0's LOW half and src1's HIGH half - in
assembler terms op_sel:[0,1(,x)] with two DIFFERENT register pairs - returns a wrong value in lanes 48..63 (the last
of the four 16-lane passes) about 6 % of the time (v_pk_fma_f32: 0.002 %) when ANOTHER wave on the same SIMD is
executing v_mfma_f32_16x16x32_f16 or v_mfma_f32_16x16x32_bf16.  Wait states around the instruction do not help (it
fails between s_nop 7 pads), every other op_sel combination measured correct (src0 swapped, both swapped, either
half broadcast from src0, src1.lo broadcast, src1 swapped with src0 == src1, src2 swapped, v_pk_mov_b32), and
v_mfma_f32_16x16x4_f32 / v_mfma_f32_16x16x16_f16 partners do not trigger it.  The LLVM of ROCm 7.2 emits the form
freely (SLP-vectorised arithmetic whose operands sit crosswise in their register pairs).

mul, add and the product of fma are commutative, and the mirrored form - op_sel:[1,0(,x)]: the low result reads
src0.HIGH and src1.LOW - is one of the forms measured correct, so the fix is a pure operand exchange on the compiler's
assembly: src0 <-> src1 together with their op_sel / op_sel_hi / neg_lo / neg_hi entries.  __graft_entry__.build()
compiles every translation unit to device assembly, runs legalize() over it, checks with audit() that no instruction
of the bad form is left, and assembles the result.

    python tools/gfx950_pk_legalize.py file.s [...]      # audit only: lists offending instructions per kernel
"""
import re
import sys

_INSTR = re.compile(r'^(\s*)(v_pk_(?:mul|add|fma)_f32)(\s+)([^;\n]*?)(\s*(?:;.*)?)$')
_MOD = re.compile(r'\s+(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]')


def _parse(line):
    m = _INSTR.match(line.rstrip('\n'))
    if not m:
        return None
    indent, op, gap, body, tail = m.groups()
    mods = {k: [int(x) for x in v.split(',')] for k, v in _MOD.findall(' ' + body)}
    operands = [o.strip() for o in _MOD.sub('', ' ' + body).split(',')]
    n_src = 3 if op == 'v_pk_fma_f32' else 2
    if len(operands) != 1 + n_src:
        raise ValueError('cannot parse packed instruction: %r' % line)
    return indent, op, gap, operands, mods, tail, n_src


def _is_bad(operands, mods, n_src):
    sel = mods.get('op_sel', [0] * n_src)
    return sel[0] == 0 and sel[1] == 1 and operands[1] != operands[2]


def legalize(text):
    """Returns (new_text, number of rewritten instructions)."""
    out, n = [], 0
    for line in text.split('\n'):
        p = _parse(line) if 'v_pk_' in line else None
        if p is None:
            out.append(line)
            continue
        indent, op, gap, operands, mods, tail, n_src = p
        if not _is_bad(operands, mods, n_src):
            out.append(line)
            continue
        operands[1], operands[2] = operands[2], operands[1]
        defaults = {'op_sel': 0, 'op_sel_hi': 1, 'neg_lo': 0, 'neg_hi': 0}
        new_mods = []
        for key in ('op_sel', 'op_sel_hi', 'neg_lo', 'neg_hi'):
            v = mods.get(key)
            if v is None:
                continue
            v[0], v[1] = v[1], v[0]
            if any(x != defaults[key] for x in v):
                new_mods.append('%s:[%s]' % (key, ','.join(str(x) for x in v)))
        out.append('%s%s%s%s%s%s' % (indent, op, gap, ', '.join(operands), ''.join(' ' + m for m in new_mods), tail))
        n += 1
    return '\n'.join(out), n


def audit(text):
    """Returns [(kernel, line number, instruction)] of every packed-fp32 instruction of the bad form, plus the count of
    packed-fp32 instructions seen (so that an empty list is known to come from a file that has any)."""
    kernel, bad, seen = None, [], 0
    for i, line in enumerate(text.split('\n'), 1):
        m = re.match(r'^(\w+):', line)
        if m and not line.startswith('.L'):
            kernel = m.group(1)
        if 'v_pk_' not in line:
            continue
        p = _parse(line)
        if p is None:
            continue
        seen += 1
        if _is_bad(p[3], p[4], p[6]):
            bad.append((kernel, i, line.strip()))
    return bad, seen


if __name__ == '__main__':
    rc = 0
    for path in sys.argv[1:]:
        bad, seen = audit(open(path).read())
        print('%s: %d packed-fp32 instructions, %d of the form gfx950 miscomputes next to a K=32 16-bit MFMA' % (path, seen, len(bad)))
        per = {}
        for k, i, ins in bad:
            per.setdefault(k, []).append((i, ins))
        for k in sorted(per):
            print('  %s: %d' % (k, len(per[k])))
            for i, ins in per[k][:4]:
                print('      line %d: %s' % (i, ins))
        rc |= 1 if bad else 0
    sys.exit(rc)
