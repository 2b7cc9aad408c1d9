"""Developer tool: one line per kernel of the -Rpass-analysis=kernel-resource-usage remarks of a library build.
    python tools/resource_usage.py [filter]        (rebuilds liblmpc_hip.so with remarks on; ~1 min)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from racinglmpc_amd import build as b

flt = sys.argv[1] if len(sys.argv) > 1 else ""
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
cmd = [hipcc, "-Rpass-analysis=kernel-resource-usage", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-Wno-unused-value",
       "-fPIC", "-shared", "-o", "/tmp/_ru.so", b.SRC, "-L/opt/rocm/lib", "-lrccl"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]; rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
    elif "error" in line:
        print(line)
print("%-62s %5s %5s %6s %6s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "sSpill", "vSpill", "scratch", "occ", "LDS"))
for k, r in rows.items():
    if flt in k:
        print("%-62s %5d %5d %6d %6d %7d %4d %7d" % (k[-62:], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("SGPRs Spill", -1), r.get("VGPRs Spill", -1),
                                                   r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("LDS Size", -1)))
