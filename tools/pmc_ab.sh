#!/bin/bash
# Developer tool (GPU box): the same counters through two developer libraries, per launch of the solve kernel -- which resource differs between two code states.
#   tools/pmc_ab.sh <tag> <libA> <libB> [batch] [horizon]
set -u
TAG=$1; LA=$2; LB=$3; B=${4:-1024}; H=${5:-40}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/${TAG}_pmc_ab.txt
for L in $LA $LB; do
  CMD="python $ROOT/bench.py --batch $B --horizon $H --steps 12 --warmup 3 --no-cpu-baseline --no-extras"
  i=1
  for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
             "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_MISSES" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_IFETCH"; do
    rm -rf /tmp/prof_ab$i
    LMPC_LIB=$ROOT/racinglmpc_amd/liblmpc_hip_$L.so timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/prof_ab$i -- $CMD > /dev/null 2>&1
    f=$(find /tmp/prof_ab$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python3 - "$f" "$L" >> $OUT/${TAG}_pmc_ab.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "lmpc_solve_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print("%s %-32s %.5g (n=%d)" % (sys.argv[2], c, sum(v) / len(v), len(v)))
PY
    else echo "$L pass $i: no counter file" >> $OUT/${TAG}_pmc_ab.txt; fi
    i=$((i+1))
  done
done
python3 - $OUT/${TAG}_pmc_ab.txt $LA $LB <<'PY'
import sys
d = {}
for l in open(sys.argv[1]):
    f = l.split()
    if len(f) >= 3 and f[0] in sys.argv[2:]:
        try: d.setdefault(f[1], {})[f[0]] = float(f[2])
        except ValueError: pass
print("%-34s %14s %14s %8s" % ("counter", sys.argv[2], sys.argv[3], "ratio"))
for c, v in sorted(d.items()):
    a, b = v.get(sys.argv[2]), v.get(sys.argv[3])
    if a is not None and b is not None:
        print("%-34s %14.5g %14.5g %8.3f" % (c, a, b, a / b if b else float("nan")))
PY
