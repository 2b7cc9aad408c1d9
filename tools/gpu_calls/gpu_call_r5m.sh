#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out
( timeout 200 python bench.py --gpus 2 --dry-run-rccl; echo "exit status $?" ) > $O/r5m_dry_run_rccl.txt 2>&1
( timeout 100 python bench.py --gpus 1 --dry-run-rccl; echo "exit status $?" ) >> $O/r5m_dry_run_rccl.txt 2>&1
( timeout 300 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "singular" 2>&1 | tail -3 ) > $O/r5m_pytest.txt 2>&1
tail -12 $O/r5m_dry_run_rccl.txt | cut -c1-700; tail -3 $O/r5m_pytest.txt
