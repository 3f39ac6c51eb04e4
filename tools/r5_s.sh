#!/bin/bash
# round 5, contact s: plan! + act! + push! of the 3-layer DQN vec-step in one launch (rlhip_dqn3_act_f32): parity + A/B (libA = before)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_s; mkdir -p $O
timeout 2000 python -m pytest tests/test_gpu_run.py tests/test_gpu_dqn3.py tests/test_gpu_abi_host.py tests/test_gpu_edges.py -m gpu -q 2>&1 | tail -8 | tee $O/tests.log
for rep in 1 2 3; do
  for v in A B; do
    echo "$v $(RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so timeout 120 python tools/dqn_fused.py 512 3 2>&1 | tail -1)"
    echo "$v $(RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so timeout 120 python tools/dqn_fused.py 32 3 2>&1 | tail -1)"
  done
done | tee $O/ab.txt
