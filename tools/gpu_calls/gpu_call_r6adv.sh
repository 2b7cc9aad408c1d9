#!/bin/bash
# round 6: hostile inputs at the C ABI (tools/adversarial_probe.py), one process per case
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1200 python tools/adversarial_probe.py > $O/r6_adversarial_probe.txt 2>&1
cat $O/r6_adversarial_probe.txt | cut -c1-320
