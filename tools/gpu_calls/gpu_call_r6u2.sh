#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -f $O/r6u2_nonfinite.txt
for w in x0nan x0inf uoldnan xlinnan ulinnan ztinf; do
  echo "== $w" >> $O/r6u2_nonfinite.txt
  timeout 90 python tools/nonfinite_probe.py $w 2>&1 | grep -v "^  File\|Extension modules\|Warning" | head -12 | cut -c1-400 >> $O/r6u2_nonfinite.txt
  echo "rc ${PIPESTATUS[0]}" >> $O/r6u2_nonfinite.txt
done
cat $O/r6u2_nonfinite.txt
