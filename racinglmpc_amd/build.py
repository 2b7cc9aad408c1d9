"""Build liblmpc_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "lmpc_capi.hip")
DEPS = [SRC, os.path.join(_HERE, "csrc", "lmpc_kernels.hip.h"), os.path.join(_HERE, "csrc", "lmpc_solve_mw.hip.h"), os.path.join(_HERE, "csrc", "lmpc_comm.hip.h"), os.path.join(os.path.dirname(_HERE), "include", "lmpc_hip.h")]
OUT = os.path.join(_HERE, "liblmpc_hip.so")


def _current():
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS)


def build_flavour(suffix, defines, verbose=False):
    """Developer builds next to the product library (e.g. the -DLMPC_TIMING flavour of tools/phase_timing.py): liblmpc_hip_<suffix>.so."""
    out = os.path.join(_HERE, "liblmpc_hip_%s.so" % suffix)
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in DEPS):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-Wno-unused-value", "-fPIC", "-shared"] + \
          ["-D" + d for d in defines] + ["-o", out, SRC, "-L/opt/rocm/lib", "-lrccl"]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=False):
    if not force and _current():
        return OUT
    import fcntl
    with open(OUT + ".lock", "w") as lock:             # several ranks may call build() at once: one compiles, the others wait
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or not _current():
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            tmp = OUT + ".tmp%d" % os.getpid()
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-Wno-unused-value", "-fPIC", "-shared", "-o", tmp, SRC,
                   "-L/opt/rocm/lib", "-lrccl"]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            subprocess.check_call(cmd)
            os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
