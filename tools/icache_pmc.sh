#!/bin/bash
# Developer tool (GPU box): instruction-cache counters of the solve kernels (one pass per counter group, --kernel-trace only).
#   tools/icache_pmc.sh <tag>     -> gpurun_out/<tag>_icache.txt
set -u
TAG=${1:-r5x}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/${TAG}_icache.txt
for CFG in "256 12" "4096 12" "1024 40"; do
  set -- $CFG; B=$1; H=$2
  CMD="python $ROOT/bench.py --batch $B --horizon $H --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
  i=1
  for PMC in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
    rm -rf /tmp/prof_ic$i
    timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/prof_ic$i -- $CMD > /dev/null 2>&1
    f=$(find /tmp/prof_ic$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python3 - "$f" "$B" "$H" >> $OUT/${TAG}_icache.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "lmpc_solve_kernel" in k or "lmpc_regress" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("B=%s N=%s %-60s %s" % (sys.argv[2], sys.argv[3], k[-60:], "  ".join("%s=%.4g (n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(d.items()))))
PY
    fi
    i=$((i+1))
  done
done
cat $OUT/${TAG}_icache.txt
