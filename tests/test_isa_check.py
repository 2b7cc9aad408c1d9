"""CPU: the build-time guard against live-range copies placed ahead of a flow block's EXEC restore (racinglmpc_amd/isa_check.py; DESIGN.md "N = 40").

* the scanner finds the block in the recorded excerpt of a faulty build (tests/golden/isa_fault_excerpt.s: lmpc_solve_kernel<40, 48, false, true>, the build
  that ran "12.4 instead of 11.0 iterations") and nothing in the excerpt of a sound one;
* every library the tree carries -- liblmpc_hip.so and the liblmpc_var_*.so variants, exactly the files the GPU tests load -- is clean.
"""
import glob
import os

import pytest

from tests import common
from racinglmpc_amd import isa_check

GOLD = os.path.join(common.ROOT, "tests", "golden")


def test_scanner_finds_the_recorded_fault():
    bad = open(os.path.join(GOLD, "isa_fault_excerpt.s")).read()
    hits = isa_check.scan(bad)
    assert len({(f, l) for f, l, _, _ in hits}) == 1
    f, l, ins, widen = hits[0]
    assert ins.startswith("v_accvgpr_write_b32") and widen.startswith("s_or_saveexec_b64"), hits[0]
    good = open(os.path.join(GOLD, "isa_sound_excerpt.s")).read()
    assert isa_check.scan(good) == []


def test_scanner_on_synthetic_blocks():
    ok = """
f:
\ts_and_saveexec_b64 s[4:5], vcc
\ts_cbranch_execz .L1
\tv_mov_b32_e32 v1, v2
.L1:
\tv_readlane_b32 s6, v255, 3
\ts_or_b64 exec, exec, s[4:5]
\tv_accvgpr_write_b32 a1, v1
\ts_endpgm
"""
    assert isa_check.scan(ok) == []
    bad = ok.replace("\tv_readlane_b32 s6, v255, 3\n", "\tv_readlane_b32 s6, v255, 3\n\tv_accvgpr_write_b32 a2, v7\n")
    assert [h[2] for h in isa_check.scan(bad)] == ["v_accvgpr_write_b32 a2, v7"]
    # a body block laid out of line (entered by execnz, closed by its own s_or) is not a flow block
    body = """
f:
\ts_and_saveexec_b64 s[4:5], vcc
\ts_cbranch_execnz .L2
.L1:
\ts_or_b64 exec, exec, s[4:5]
\ts_endpgm
.L2:
\tv_mov_b32_e32 v1, v2
\ts_or_b64 exec, exec, s[6:7]
\ts_branch .L1
"""
    assert isa_check.scan(body) == []


def test_libraries_in_the_tree_are_clean():
    libs = sorted(glob.glob(os.path.join(common.ROOT, "racinglmpc_amd", "liblmpc_hip.so")) + glob.glob(os.path.join(common.ROOT, "racinglmpc_amd", "liblmpc_var_N*.so")))
    if not libs or not os.path.exists(isa_check.OBJDUMP):
        pytest.skip("no built library / no llvm-objdump here")
    for lib in libs:
        hits = isa_check.check(lib)
        assert not hits, (os.path.basename(lib), sorted({(f, l) for f, l, _, _ in hits}))


def test_check_fails_closed(tmp_path):
    """a library the scanner cannot read is an error, never a clean result."""
    empty = tmp_path / "no_bundle.so"
    empty.write_bytes(b"\x7fELF" + b"\0" * 256)                              # no offload bundle inside
    if os.path.exists(isa_check.OBJDUMP):
        with pytest.raises(isa_check.IsaCheckError, match="no amdgcn code object"):
            isa_check.check(str(empty))
    blank = tmp_path / "blank.s"
    blank.write_text("f:\n")                                                # a listing with no instruction
    with pytest.raises(isa_check.IsaCheckError, match="not understood"):
        isa_check.check(str(blank))
    noexecz = tmp_path / "noexecz.s"
    noexecz.write_text("f:\n\tv_mov_b32_e32 v1, v2\n\ts_endpgm\n")           # instructions, but not one exec-mask region: not one of these kernels
    with pytest.raises(isa_check.IsaCheckError, match="not understood"):
        isa_check.check(str(noexecz))
    hits, images = isa_check.check_report(os.path.join(GOLD, "isa_sound_excerpt.s"))
    assert hits == [] and images[0]["instructions"] > 0 and images[0]["execz_branches"] > 0
    libs = sorted(glob.glob(os.path.join(common.ROOT, "racinglmpc_amd", "liblmpc_hip.so")))
    if libs and os.path.exists(isa_check.OBJDUMP):
        hits, images = isa_check.check_report(libs[0])
        assert hits == [] and images and all(i["instructions"] > 10000 and i["execz_branches"] > 10 for i in images), images


def test_pre_fix_sources_reproduce_the_fault():
    """tools/isa_fault_repro.py: the sources of the last commit before the round-5 change, compiled with the one define rounds 3-4 could not explain at N = 40
    (the 4x4x4 Gram matrix in the one-wave kernel), are flagged by the scanner -- v_accvgpr_write copies ahead of an s_or_saveexec_b64; HEAD's libraries are
    clean (test above).  Needs this repository's history and hipcc (about a minute of cross-compilation); skipped where either is missing."""
    import shutil
    import subprocess
    import sys
    if not os.path.isdir(os.path.join(common.ROOT, ".git")) or not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no git history or no hipcc here")
    if subprocess.run(["git", "-C", common.ROOT, "cat-file", "-e", "5119d31^{commit}"], capture_output=True).returncode != 0:
        pytest.skip("commit 5119d31 is not in this clone")
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "tools", "isa_fault_repro.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 1, r.stdout + r.stderr                       # (exit code 1: the scanner found blocks)
    assert "v_accvgpr_write_b32" in r.stdout and "s_or_saveexec_b64" in r.stdout and "lmpc_solve_kernelILi40" in r.stdout, r.stdout
