"""Vector-register liveness of one kernel in a gfx950 device assembly listing (hipcc -S --cuda-device-only).

  python tools/vgpr_liveness.py /tmp/nfi_kernels.s <mangled-name-substring> [--at N] [--top K]

Builds the control-flow graph from labels and branches, runs the usual backward dataflow on v0..v511 (AGPRs are not
counted), prints the pressure profile (maximum, the line where it is reached, pressure at every buffer_load cluster and
at every MFMA block) and, for the point of maximum pressure (or --at LINE), the live registers grouped by the line that
defined them - which, with the source comments hipcc leaves in the listing, says WHAT is being held across the gather.
A reading aid for the "what keeps the render kernel from a third wave per SIMD" question (HISTORY.md section 8); not part
of the build.
"""
import re
import sys

src, key = sys.argv[1], sys.argv[2]
at = int(sys.argv[sys.argv.index('--at') + 1]) if '--at' in sys.argv else None
top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 12

lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start + 1:end]

REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


# instructions whose FIRST operand is not a destination
NO_DST = ('buffer_store', 'global_store', 'flat_store', 'ds_write', 'ds_store', 'scratch_store', 's_', 'v_cmp', 'v_cmpx',
          'global_atomic', 'buffer_atomic', 'ds_add', 'ds_max', 'ds_min', 'exp', 'buffer_wbl2', 'buffer_inv')
TWO_DST = ('v_mad_u64_u32', 'v_mad_i64_i32', 'v_add_co_u32', 'v_sub_co_u32', 'v_addc_co_u32', 'v_subb_co_u32',
           'v_subrev_co_u32', 'v_div_scale', 'v_permlane32_swap', 'v_permlane16_swap', 'v_swap_b32')
# destination is also read (accumulate in place / partial write)
RMW = ('v_fmac', 'v_pk_fmac', 'v_mac', 'v_dot2c', 'v_dot4c', 'v_writelane', 'v_movrel', 'v_permlane32_swap',
       'v_permlane16_swap', 'v_swap_b32', 'v_mov_b32_dpp', 'v_cndmask_b32_dpp')

ins = []          # (line_no, text, defs, uses, mnemonic)
label_at = {}
for n, l in enumerate(body):
    m = re.match(r'^([.\w$]+):', l)
    if m:
        label_at[m.group(1)] = len(ins)
        continue
    t = l.split(';')[0].strip()
    if not t or t.startswith('.'):
        continue
    mn = t.split()[0]
    ops = t[len(mn):].strip()
    parts = [p.strip() for p in re.split(r',(?![^\[]*\])', ops)] if ops else []
    defs, uses = [], []
    if mn.startswith(NO_DST) and not mn.startswith(('s_waitcnt', 's_nop')):
        ndst = 0
        if mn.startswith(('global_atomic', 'buffer_atomic', 'ds_add', 'ds_max', 'ds_min')) and ('glc' in t or 'sc0' in t or '_rtn' in mn):
            ndst = 1
    else:
        ndst = 2 if mn.startswith(TWO_DST) and not mn.startswith(('v_permlane', 'v_swap')) else 1
    if mn.startswith(('v_permlane32_swap', 'v_permlane16_swap', 'v_swap_b32')):
        ndst = 2
    for i, p in enumerate(parts):
        (defs if i < ndst else uses).extend(regs(p))
    dpp_or_sdwa = 'dpp' in t or 'row_' in t or 'quad_perm' in t or 'sdwa' in mn or 'dst_sel' in t
    if mn.startswith(RMW) or (dpp_or_sdwa and mn.startswith('v_')) or 'op_sel' in t and mn.startswith('v_fma_mix'):
        uses.extend(defs[:])          # partial / accumulating writes keep the old value alive
    if mn.startswith(('v_mfma', 'v_smfmac')) and len(parts) >= 4:
        pass                           # C operand is an ordinary use (already in uses)
    if 'lds' in t.split() and mn.startswith('buffer_load'):
        defs = []
    ins.append((start + 2 + n, t, set(defs), set(uses), mn))

N = len(ins)
succ = [[] for _ in range(N)]
for i, (_, t, _, _, mn) in enumerate(ins):
    if mn == 's_endpgm':
        continue
    if mn == 's_branch':
        succ[i].append(label_at[t.split()[1]])
        continue
    if mn.startswith('s_cbranch'):
        succ[i].append(label_at[t.split()[1]])
    if i + 1 < N:
        succ[i].append(i + 1)

live_in = [set() for _ in range(N)]
changed = True
while changed:
    changed = False
    for i in range(N - 1, -1, -1):
        out = set()
        for s in succ[i]:
            out |= live_in[s]
        new = (out - ins[i][2]) | ins[i][3]
        if new != live_in[i]:
            live_in[i] = new
            changed = True

pressure = [len(s) for s in live_in]
peak = max(range(N), key=lambda i: pressure[i])
print('kernel at line %d: %d instructions, highest register named v%d' %
      (start + 1, N, max([max(d) for _, _, d, _, _ in ins if d] + [0])))
print('maximum live vector registers: %d at line %d: %s' % (pressure[peak], ins[peak][0], ins[peak][1][:80]))

# pressure at load clusters and MFMA blocks
print('\nregion                      first line  live-in  (first instruction)')
prev = None
for i, (ln, t, _, _, mn) in enumerate(ins):
    kind = 'gather' if mn.startswith('buffer_load_dwordx4') else 'mfma' if mn.startswith('v_mfma') else None
    if kind and kind != prev:
        print('%-27s %10d  %7d  %s' % (kind, ln, pressure[i], t[:60]))
    if kind or not mn.startswith(('s_', 'v_')):
        prev = kind if kind else prev
    if mn.startswith(('s_cbranch', 's_branch', 's_barrier')):
        prev = None

# who is live at the chosen point
idx = peak if at is None else min(range(N), key=lambda i: abs(ins[i][0] - at))
live = live_in[idx]
print('\nlive at line %d (%d registers), grouped by the defining instruction nearest above:' % (ins[idx][0], len(live)))
groups = {}
for r in sorted(live):
    j = idx - 1
    while j >= 0 and r not in ins[j][2]:
        j -= 1
    key2 = (ins[j][0], ins[j][1][:70]) if j >= 0 else (0, '(kernel entry / loop-carried from below)')
    groups.setdefault(key2, []).append(r)
for (ln, t), rs in sorted(groups.items(), key=lambda kv: -len(kv[1]))[:top * 4]:
    print('  %3d  line %-8d %s' % (len(rs), ln, t))
