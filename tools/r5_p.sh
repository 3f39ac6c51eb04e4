#!/bin/bash
# round 5, contact p: dqn3_grad32 with the db3 / loss sums deferred behind the tile loop: parity + same-box A/B (libA = before, libB = after)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_p; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dqn3.py tests/test_gpu_run.py tests/test_gpu_abi_host.py -m gpu -q 2>&1 | tail -6 | tee $O/tests.log
for rep in 1 2 3; do
  for v in A B; do
    for b in 131072 4096; do echo "$v batch $b: $(RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so timeout 120 python tools/dqn3w_time.py 128 $b 2>&1 | tail -1)"; done
    echo "$v $(RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so timeout 120 python tools/dqn_fused.py 512 3 2>&1 | tail -1)"
  done
done | tee $O/ab.txt
