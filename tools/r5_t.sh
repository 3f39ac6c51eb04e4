#!/bin/bash
# round 5, contact t: plan! / act! of the 2-layer DQN on 16-lane rows with the network staged in LDS: parity + A/B (libA = before)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_t; mkdir -p $O
timeout 2000 python -m pytest tests/test_gpu_run.py tests/test_gpu_learners.py tests/test_gpu_abi_host.py tests/test_gpu_explorers.py tests/test_gpu_bench_shapes.py -m gpu -q 2>&1 | tail -8 | tee $O/tests.log
for rep in 1 2 3; do
  for v in A B; do
    for b in 32 512 4096; do echo "$v $(RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so timeout 120 python tools/dqn_fused.py $b 2 2>&1 | tail -1)"; done
  done
done | tee $O/ab.txt
