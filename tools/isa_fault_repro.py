"""Developer tool (CPU, cross-compile only): reproduce the compiler fault behind rounds 3-4's "unexplained" N = 40 builds.

    python tools/isa_fault_repro.py [commit]          # default: 5119d31, the last commit before the sources were changed around the fault

The sources of that commit (racinglmpc_amd/csrc, from this repository's own history) are compiled for gfx950 with the one define that rounds 3-4 could
not explain at N = 40 (-DLMPC_FORCE_GRAM8: the 4x4x4-MFMA Gram matrix of the terminal factor in the one-wave kernel), only the N = 40 kernels
(-DLMPC_DEV_FAST -DLMPC_DEV_N=40), with the product's options, and racinglmpc_amd/isa_check.py scans the result: it prints the flow blocks in which
ROCm 7.2's register allocator put per-lane copies (v_accvgpr_write / v_mov of loop-invariant registers) AHEAD of the block's s_or_saveexec_b64, i.e. under
the THEN mask only -- lanes 48..63 of the selection-cost / affine-term registers then hold garbage, the first pass ends LMPC_ST_NUMERIC on every problem
and the retry kernel hides it (the "12.4 iterations, every certificate green" of round 4).  The same command on HEAD reports nothing: the construct
(a per-lane if / else around register arrays that stay live across it) is gone from the one-wave kernel, and build.py refuses any build the scan flags.
This is a reproducer from history, not a stand-alone 60-line one: the fault needs the kernel's register pressure (512 registers, ~40 spilled) to
make the allocator split live ranges at that block, and every reduced kernel tried allocated without splits.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from racinglmpc_amd import isa_check                                     # noqa: E402

commit = sys.argv[1] if len(sys.argv) > 1 else "5119d31"
work = os.path.join(ROOT, "build_tmp", "isa_repro_" + commit)
os.makedirs(os.path.join(work, "include"), exist_ok=True)
for f in subprocess.check_output(["git", "-C", ROOT, "ls-tree", "-r", "--name-only", commit, "racinglmpc_amd/csrc", "include"]).decode().split():
    dst = os.path.join(work, "include" if f.startswith("include/") else "", os.path.basename(f))
    open(dst, "wb").write(subprocess.check_output(["git", "-C", ROOT, "show", "%s:%s" % (commit, f)]))
out = os.path.join(work, "liblmpc_hip_gram40.so")
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-Wno-unused-value",
                       "-fPIC", "-shared", "-DLMPC_DEV_FAST", "-DLMPC_DEV_N=40", "-DLMPC_FORCE_GRAM8", "-I" + os.path.join(work, "include"), os.path.join(work, "lmpc_capi.hip"),
                       "-L/opt/rocm/lib", "-lrccl", "-o", out])
hits = isa_check.check(out)
print("%s at %s: %d per-lane instruction(s) ahead of an EXEC restore" % (os.path.basename(out), commit, len(hits)))
for fn, label, line, ins in hits[:12]:
    print("   %s  block %s: `%s` runs before `%s`" % (fn[:70], label, line, ins))
sys.exit(1 if hits else 0)
