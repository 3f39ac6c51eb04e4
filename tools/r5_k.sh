#!/bin/bash
# round 5, contact k: the one-launch DQN optimise! (dqn_grad_kernel<..., FUSE>): parity subset + vec-step timing, fused vs RLHIP_DQN_NO_FUSE=1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_learners.py tests/test_gpu_run.py tests/test_gpu_abi_host.py -m gpu -q -k "dqn or DQN or fused or host" 2>&1 | tail -15 | tee $O/tests.log
for rep in 1 2 3; do
  for b in 32 512 4096; do
    RLHIP_DQN_NO_FUSE=1 timeout 120 python tools/dqn_fused.py $b 2 2>&1 | tail -1 | sed 's/^/two-launch optimise: /'
    timeout 120 python tools/dqn_fused.py $b 2 2>&1 | tail -1 | sed 's/^/one-launch optimise: /'
  done
done | tee $O/ab.txt
