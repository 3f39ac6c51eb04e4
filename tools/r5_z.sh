#!/bin/bash
# round 5, contact z2: write-through partial rows in the DQN learners: vec-step A/B (libA = plain stores, HEAD before the change)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_z; mkdir -p $O
for rep in 1 2 3; do
  for v in A B; do
    export RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so
    for b in 512 4096; do echo "$v $(timeout 120 python tools/dqn_fused.py $b 2 2>&1 | tail -1)"; done
    for b in 512 4096; do echo "$v $(timeout 120 python tools/dqn_fused.py $b 3 2>&1 | tail -1)"; done
  done
done | tee $O/ab2.txt
