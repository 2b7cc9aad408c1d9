#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -f $O/r6u2_nonfinite.txt
for w in ztinf x0nan; do
  echo "== $w" >> $O/r6u2_nonfinite.txt
  timeout 90 python tools/nonfinite_probe.py $w 2>&1 | grep -v "^  File\|Extension modules\|Warning" | head -12 | cut -c1-400 >> $O/r6u2_nonfinite.txt
  echo "rc ${PIPESTATUS[0]}" >> $O/r6u2_nonfinite.txt
done
cat $O/r6u2_nonfinite.txt
( timeout 1000 python -m pytest tests/ -x -q -m gpu -rA -s -p no:cacheprovider 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn" ) > $O/r6_final_pytest.txt 2>&1
grep -n "passed\|failed" $O/r6_final_pytest.txt | tail -2; grep -n "FAILED\|poisoned" $O/r6_final_pytest.txt | head
