#!/bin/bash
# round 6: kernel trace + PMC passes of the two-wave route (batch 1024, N = 12) on the final kernels -- the per-batch loop of tools/collect_profiles.sh under the r6zz tag
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; TAG=r6zz; B=1024
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --batch $B --horizon 12 --steps 30 --warmup 5 --no-cpu-baseline --no-extras"
rm -rf /tmp/prof_stats
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- $CMD > $OUT/${TAG}_B${B}_bench_line.json 2> /dev/null
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_B${B}_kernel_stats.csv
i=1
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_F64"; do
  rm -rf /tmp/prof_pmc$i
  timeout 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/prof_pmc$i -- $CMD > /dev/null 2>&1
  f=$(find /tmp/prof_pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then (head -1 $f; grep -E "lmpc_solve_kernel|lmpc_regress_kernel" $f) > $OUT/${TAG}_B${B}_pmc_pass$i.csv; fi
  i=$((i+1))
done
ls -la $OUT | grep ${TAG}_B${B}; head -4 $OUT/${TAG}_B${B}_kernel_stats.csv | cut -c1-200
