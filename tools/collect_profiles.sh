#!/bin/bash
# Collect the measurement set of one round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r2a [batch ...]
# writes gpurun_out/<tag>_bench.json (the driver's command: every configuration in one line) and, per batch size B (default 256 4096),
# gpurun_out/<tag>_B<B>_kernel_stats.csv and _B<B>_pmc_pass{1..6}.csv.  Copy them to profiles/ and run
# tools/make_traffic.py profiles/<tag> to refresh profiles/traffic.json (read by bench.py's roofline object).
set -u
TAG=${1:-r2x}
shift
BATCHES=${@:-256 4096}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
H=${HORIZON:-12}                                   # HORIZON=40 tools/collect_profiles.sh r3m_N40 1024: kernel stats + PMC passes of another horizon only
if [ "$H" = "12" ]; then python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; fi
cd /tmp && export TMPDIR=/tmp
for B in $BATCHES; do
  CMD="python $ROOT/bench.py --batch $B --horizon $H --steps 30 --warmup 5 --no-cpu-baseline --no-extras"
  rm -rf /tmp/prof_stats
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- $CMD > /dev/null 2>&1
  cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_B${B}_kernel_stats.csv
  i=1
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"; do
    rm -rf /tmp/prof_pmc$i
    rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/prof_pmc$i -- $CMD > /dev/null 2>&1
    f=$(find /tmp/prof_pmc$i -name "*counter_collection.csv" | head -1)
    # keep the two hot kernels only (the files are large otherwise)
    if [ -n "$f" ]; then (head -1 $f; grep -E "lmpc_solve_kernel|lmpc_regress_kernel|lmpc_step_kernel" $f) > $OUT/${TAG}_B${B}_pmc_pass$i.csv; fi
    i=$((i+1))
  done
done
cd $ROOT
ls -la $OUT | grep $TAG
if [ "$H" != "12" ]; then exit 0; fi
# phase timings of one QP (developer build with cycle stamps) and the RCCL branch on one rank
LMPC_TIMING_MW=1 python tools/phase_timing.py > $OUT/${TAG}_phase_timing_mw4.txt 2>&1
python tools/phase_timing.py > $OUT/${TAG}_phase_timing_1w.txt 2>&1
LMPC_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rollouts-per-gpu 256 > $OUT/${TAG}_bench_rccl_1rank.json 2> $OUT/${TAG}_bench_rccl_1rank.err
