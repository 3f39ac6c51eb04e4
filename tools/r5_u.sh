#!/bin/bash
# round 5, contact u: dqn_reduce_apply_kernel's Float64 block sum on DPP: parity + A/B at the batches that use it (> 2048 samples)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_u; mkdir -p $O
timeout 1000 python -m pytest tests/test_gpu_learners.py -m gpu -q -k "dqn" 2>&1 | tail -4 | tee $O/tests.log
for rep in 1 2 3; do
  for v in A B; do
    for b in 4096 8192; do echo "$v $(RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so timeout 120 python tools/dqn_fused.py $b 2 2>&1 | tail -1)"; done
  done
done | tee $O/ab.txt
