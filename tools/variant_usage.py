"""Developer tool: kernel-resource-usage remarks of ONE (N, numSS_points) variant translation unit, one line per kernel (seconds, not the whole library).
    python tools/variant_usage.py N S [extra hipcc flags]"""
import re
import subprocess
import sys

N, S = sys.argv[1], sys.argv[2]
cmd = ["/opt/rocm/bin/hipcc", "-Rpass-analysis=kernel-resource-usage", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-Wno-unused-value",
       "-DLMPC_VARIANT_TU", "-DLMPC_VAR_N=" + N, "-DLMPC_VAR_S=" + S, "--cuda-device-only", "-c", "-o", "/tmp/_vu.o", "racinglmpc_amd/csrc/lmpc_variant.hip"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]; rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print("%-52s %5s %5s %6s %6s %7s %4s %7s   (loop-not-unrolled warnings: %d)" % ("kernel", "VGPR", "AGPR", "sSpill", "vSpill", "scratch", "occ", "LDS", out.count("loop not unrolled")))
for k, r in rows.items():
    print("%-52s %5d %5d %6d %6d %7d %4d %7d" % (k[-52:], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("SGPRs Spill", -1), r.get("VGPRs Spill", -1), r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("LDS Size", -1)))
