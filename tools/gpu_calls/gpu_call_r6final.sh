#!/bin/bash
# round 6, final verification as the driver runs it: GPU suite, smoke(), default bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( timeout 1000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 ) > $O/r6_final_pytest.txt 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/r6_final_smoke.txt 2>&1
( time timeout 600 python bench.py ) > $O/r6_final_bench.json 2> $O/r6_final_bench.err
cat $O/r6_final_pytest.txt; cat $O/r6_final_smoke.txt | cut -c1-300; python tools/show_bench.py $O/r6_final_bench.json | head -3 | cut -c1-300; tail -4 $O/r6_final_bench.err
