#!/bin/bash
# round 6, the default bench and smoke() as the driver runs them (after the last measurement set: profiles/traffic.json stamp)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/r6_final_smoke.txt 2>&1
( time timeout 600 python bench.py ) > $O/r6_final_bench.json 2> $O/r6_final_bench.err
cat $O/r6_final_smoke.txt | cut -c1-300; python tools/show_bench.py $O/r6_final_bench.json | head -3 | cut -c1-300; tail -4 $O/r6_final_bench.err
