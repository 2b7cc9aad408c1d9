#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 500 python tools/n40_experiments.py run ) > $O/r5d_n40.txt 2>&1
grep "^N40" $O/r5d_n40.txt | cut -c1-420
