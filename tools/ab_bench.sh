#!/bin/bash
# Developer tool: A/B throughput table of two builds of the library on the GPU box (tools/exp_bench.py per build).
#   tools/ab_bench.sh <tag> <libA.so> <libB.so> [batch sizes ...]
TAG=$1; A=$2; B=$3; shift 3
SIZES=${@:-1 64 256 512 1024 4096 8192}
mkdir -p gpurun_out
for L in $A $B; do
  echo "== $L" >> gpurun_out/${TAG}_ab.txt
  LMPC_LIB=$(pwd)/$L EXP_CERT=1 timeout 300 python tools/exp_bench.py $SIZES >> gpurun_out/${TAG}_ab.txt 2>&1
done
cat gpurun_out/${TAG}_ab.txt
