"""tools/isa_lint.py, the build's check of the compiler's assembly: it must report lane-masked code ahead of an
EXEC restore in a join block, and nothing else that merely looks like it."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LINT = ROOT / "tools" / "isa_lint.py"

BAD = """
kern:
\ts_and_saveexec_b64 s[6:7], vcc
\ts_cbranch_execz .LBB0_2
; %bb.1:
\tv_add_u32_e32 v1, v1, v2
.LBB0_2:
\tv_writelane_b32 v95, s54, 28
\tv_mov_b32_e32 v43, v52
\ts_or_b64 exec, exec, s[6:7]
\ts_swappc_b64 s[30:31], s[0:1]
\tv_mov_b32_e32 v52, v43
\ts_endpgm
.Lfunc_end0:
\t.amdhsa_kernel kern
\t\t.amdhsa_private_segment_fixed_size 40
\t.end_amdhsa_kernel
"""

# the same shapes where they are legitimate: spill lanes only ahead of the restore; a conditional block laid out of
# line (entered through execnz only) with the join's first instructions copied behind it
GOOD = """
kern:
\ts_and_saveexec_b64 s[6:7], vcc
\ts_cbranch_execz .LBB0_2
; %bb.1:
\tv_add_u32_e32 v1, v1, v2
.LBB0_2:
\tv_writelane_b32 v95, s54, 28
\ts_or_b64 exec, exec, s[6:7]
\tv_mov_b32_e32 v43, v52
\ts_and_saveexec_b64 s[0:1], s[20:21]
\ts_cbranch_execnz .LBB0_5
.LBB0_3:
\ts_or_b64 exec, exec, s[0:1]
\ts_endpgm
.LBB0_5:
\tv_mul_lo_u32 v5, v5, 20
\tglobal_store_byte v[22:23], v5, off
\ts_or_b64 exec, exec, s[0:1]
\ts_endpgm
.Lfunc_end0:
\t.amdhsa_kernel kern
\t\t.amdhsa_private_segment_fixed_size 40
\t.end_amdhsa_kernel
"""


def run(tmp_path, text, *args):
    f = tmp_path / "k.s"
    f.write_text(text)
    return subprocess.run([sys.executable, str(LINT), str(f), *args], capture_output=True, text=True)


def test_masked_copy_ahead_of_the_restore_is_reported(tmp_path):
    r = run(tmp_path, BAD)
    assert r.returncode == 1
    assert "v_mov_b32_e32 v43, v52" in r.stderr and "v_writelane" not in r.stderr


def test_legitimate_layouts_pass(tmp_path):
    r = run(tmp_path, GOOD)
    assert r.returncode == 0, r.stderr


def test_scratch_budget(tmp_path):
    assert run(tmp_path, GOOD, "--max-scratch", "kern=64").returncode == 0
    r = run(tmp_path, GOOD, "--max-scratch", "kern=32")
    assert r.returncode == 1 and "40 bytes of scratch" in r.stderr
