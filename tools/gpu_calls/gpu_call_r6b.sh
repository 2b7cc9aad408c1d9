#!/bin/bash
# round 6: capture the closed-loop QPs on which the kernel and the oracle's dense interior-point optimum disagree by more than 1e-7 (both routes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for s in 6 7; do timeout 600 python tools/capture_probe_misses.py dropin $s 12 2>&1 | tail -2; done
for s in 5 6; do timeout 600 python tools/capture_probe_misses.py rollouts $s 12 2>&1 | tail -2; done
