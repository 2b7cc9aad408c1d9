#!/bin/bash
# Collect the measurement set of one round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r1g
# writes gpurun_out/<tag>_bench.json, _bench_sweep.json, _kernel_stats.csv, _pmc_pass{1..6}.csv; copy them to profiles/ and run
# tools/make_traffic.py profiles/<tag> to refresh profiles/traffic.json (read by bench.py's roofline object).
set -u
TAG=${1:-r1x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python bench.py --sweep --no-cpu-baseline > $OUT/${TAG}_bench_sweep.json 2>> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- $CMD > /dev/null 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
i=1
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"; do
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/prof_pmc$i -- $CMD > /dev/null 2>&1
  f=$(find /tmp/prof_pmc$i -name "*counter_collection.csv" | head -1)
  # keep the two hot kernels only (the files are large otherwise)
  if [ -n "$f" ]; then (head -1 $f; grep -E "lmpc_solve_kernel|lmpc_regress_kernel" $f) > $OUT/${TAG}_pmc_pass$i.csv; fi
  i=$((i+1))
done
cd $ROOT
ls -la $OUT | grep $TAG
