"""tools/make_traffic.py profiles/<tag>_B<batch> [...]  ->  profiles/traffic.json   (keys carry the batch size parsed from the tag)

HBM bytes per launch of the two hot kernels from the rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are collected in separate
passes, tools/collect_profiles.sh), corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950:
bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024 (FETCH_SIZE reports half of wide coalesced reads)."""
import csv, json, sys, collections

import re, os
tag = sys.argv[1]
BK = "B%s_N%s" % (re.search(r"_B(\d+)$", tag).group(1) if re.search(r"_B(\d+)$", tag) else "256",
                  re.search(r"_N(\d+)_B", tag).group(1) if re.search(r"_N(\d+)_B", tag) else "12")


def mean_counter(path, name):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == name:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = mean_counter(tag + "_pmc_pass1.csv", "FETCH_SIZE")
write = mean_counter(tag + "_pmc_pass2.csv", "WRITE_SIZE")
out = {}
if os.path.exists("profiles/traffic.json") and os.environ.get("TRAFFIC_MERGE", "1") == "1":
    out = json.load(open("profiles/traffic.json"))
def short_name(kname):
    m = re.search(r"lmpc_solve_kernel<([^>]*)>", kname)                          # the retry variant lmpc_solve_kernel<N, S, true[, false]>: not a bench kernel
    if m and len(m.group(1).split(",")) >= 3 and m.group(1).split(",")[2].strip() == "true":
        return None
    return "lmpc_solve_kernel" if "lmpc_solve_kernel" in kname else ("lmpc_regress_kernel" if "lmpc_regress_kernel" in kname else None)


for kname in fetch:
    short = short_name(kname)
    if short is None:
        continue
    out[short + "_bytes_per_launch_" + BK] = 2 * fetch[kname] * 1024 + write.get(kname, 0.0) * 1024
    out[short + "_raw_" + BK] = {"kernel": kname[:60], "FETCH_SIZE_KB": fetch[kname], "WRITE_SIZE_KB": write.get(kname, 0.0)}
# instruction mix and utilisation of the solve kernel (passes 3-6), per launch
def all_counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        with open(path) as f:
            for row in csv.DictReader(f):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    except FileNotFoundError:
        pass
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


cnt = collections.defaultdict(dict)
for i in (3, 4, 5, 6):
    for kname, d in all_counters("%s_pmc_pass%d.csv" % (tag, i)).items():
        short = short_name(kname)
        if short:
            cnt[short].update(d)
for short, d in cnt.items():
    if "SQ_INSTS_VALU_FMA_F64" in d:
        # FP64 flop per launch: 64 lanes x (2 per FMA + 1 per add / mul) + 512 per 4x4x4 (4 blocks) or 2048 per 16x16x4 MFMA -- counted as 512 (lower bound)
        d["fp64_flop_per_launch"] = 64.0 * (2 * d["SQ_INSTS_VALU_FMA_F64"] + d.get("SQ_INSTS_VALU_ADD_F64", 0) + d.get("SQ_INSTS_VALU_MUL_F64", 0)) + 512.0 * d.get("SQ_INSTS_VALU_MFMA_F64", 0)
    if "SQ_WAVE_CYCLES" in d and "SQ_ACTIVE_INST_VALU" in d:
        d["valu_utilisation"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]
    if "SQ_LDS_IDX_ACTIVE" in d and d["SQ_LDS_IDX_ACTIVE"] > 0:
        d["lds_bank_conflict_rate"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
    out[short + "_counters_" + BK] = d
out["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, %s_pmc_pass1/2.csv), mean over the bench launches, N=12; "
               "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (MI355X_MICROARCH.md HBM section: FETCH_SIZE reports 1/2 of wide coalesced reads on gfx950; "
               "WRITE_SIZE uncalibrated; Infinity-Cache hits are included)." % tag)
import subprocess
try:      # the commit the counters belong to (collected from a snapshot of the working tree: HEAD at collection time, "+" if it was dirty)
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "status", "--porcelain", "racinglmpc_amd/csrc"], capture_output=True, text=True).stdout.strip() != ""
    out["commit"] = os.environ.get("TRAFFIC_COMMIT") or (head + ("+" if dirty else ""))
    out["collected_with_tag"] = os.path.basename(tag).rsplit("_B", 1)[0].replace("_N40", "")
except Exception:
    pass
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
