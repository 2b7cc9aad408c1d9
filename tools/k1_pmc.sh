#!/bin/bash
# Developer tool: counters of the regression kernel at one batch size (run through gpurun from the repo root):  tools/k1_pmc.sh <out tag> [batch]
TAG=${1:-k1}; B=${2:-4096}; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --batch $B --steps 20 --warmup 3 --no-cpu-baseline --no-extras"
i=1
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"; do
  rm -rf /tmp/k1pmc$i
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/k1pmc$i -- $CMD > /dev/null 2>&1
  f=$(find /tmp/k1pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then (head -1 $f; grep -E "lmpc_regress_kernel" $f | head -400) > $OUT/${TAG}_B${B}_k1pmc$i.csv; fi
  i=$((i+1))
done
