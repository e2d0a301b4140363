"""CPU test of tools/vgpr_liveness.py (the register-liveness reader used for the occupancy work, HISTORY.md section 7): a
hand-written listing with a loop, a loop-carried value, a value held across a load cluster and a dead definition."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LISTING = """\t.text
_Z4demoPf:                              ; @_Z4demoPf
\tv_mov_b32_e32 v0, 1.0
\tv_mov_b32_e32 v9, 0
\tv_mov_b32_e32 v7, 2.0
.LBB0_1:
\tbuffer_load_dwordx4 v[1:4], v9, s[0:3], 0 offen
\tv_mov_b32_e32 v8, 3.0
\ts_waitcnt vmcnt(0)
\tv_pk_fma_f32 v[5:6], v[1:2], v[3:4], v[5:6]
\tv_fmac_f32_e32 v0, v5, v7
\tv_add_u32_e32 v9, 16, v9
\ts_cbranch_scc1 .LBB0_1
\tglobal_store_dword v9, v0, s[4:5]
\ts_endpgm
.Lfunc_end0:
"""


def test_liveness_of_a_small_loop(tmp_path):
    src = tmp_path / 'demo.s'
    src.write_text(LISTING)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'vgpr_liveness.py'), str(src), 'demo'],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    # at the wait: the four loaded registers, the accumulator pair v[5:6] (read-modify-write across the back edge), the
    # loop-carried v0 / v9 and the loop-invariant v7 are live; v8 is written and never read, so it never counts
    assert 'maximum live vector registers: 9' in out, out
    assert 'gather' in out                         # the load cluster is reported with its live-in count
    peak = out[out.index('live at line'):]
    assert 'buffer_load_dwordx4 v[1:4]' in peak and 'v_mov_b32_e32 v7, 2.0' in peak
    assert 'v_mov_b32_e32 v8' not in peak
