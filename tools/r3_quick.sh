#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/quick; mkdir -p $O
timeout 600 python tools/diag_tanh.py 2>&1 | grep -v amdgpu.ids | tee $O/diag.txt
timeout 900 python -m pytest tests/test_gpu_ppo3w.py -q 2>&1 | tail -15 | tee $O/tests.txt
