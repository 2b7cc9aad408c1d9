"""tools/make_traffic.py profiles/<tag>  ->  profiles/traffic.json

HBM bytes per launch of the two hot kernels from the rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are collected in separate
passes, tools/collect_profiles.sh), corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950:
bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024 (FETCH_SIZE reports half of wide coalesced reads)."""
import csv, json, sys, collections

tag = sys.argv[1]


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
for kname in fetch:
    short = "lmpc_solve_kernel" if "lmpc_solve_kernel" in kname else ("lmpc_regress_kernel" if "lmpc_regress_kernel" in kname else None)
    if short is None:
        continue
    out[short + "_bytes_per_launch_B256_N12"] = 2 * fetch[kname] * 1024 + write.get(kname, 0.0) * 1024
    out[short + "_raw"] = {"kernel": kname[:60], "FETCH_SIZE_KB": fetch[kname], "WRITE_SIZE_KB": write.get(kname, 0.0)}
out["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, %s_pmc_pass1/2.csv), mean over the bench launches at B=256, N=12; "
               "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (MI355X_MICROARCH.md HBM section: FETCH_SIZE reports 1/2 of wide coalesced reads on gfx950; "
               "WRITE_SIZE uncalibrated; Infinity-Cache hits are included)." % tag)
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
