#!/bin/bash
# round 5, contact r: block sums of the learner tails on DPP (block_sum_f64_dpp): parity of the 3-layer learners + same-box A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_r; mkdir -p $O
timeout 2000 python -m pytest tests/test_gpu_dqn3.py tests/test_gpu_dqn3w.py tests/test_gpu_ppo3w.py tests/test_gpu_ppo3.py tests/test_gpu_run.py -m gpu -q 2>&1 | tail -6 | tee $O/tests.log
for rep in 1 2 3; do
  for v in A B; do
    echo "$v $(RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so timeout 120 python tools/dqn_fused.py 512 3 2>&1 | tail -1)"
    echo "$v $(RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so timeout 120 python tools/dqn_fused.py 4096 3 2>&1 | tail -1)"
  done
done | tee $O/ab.txt
