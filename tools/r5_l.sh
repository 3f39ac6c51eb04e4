#!/bin/bash
# round 5, contact l: stamped timeline of the fused DQN vec-step (timing build gpurun_ab/libT.so)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_l; mkdir -p $O
for b in 512 512 32 4096; do RLHIP_LIB_PATH=$PWD/gpurun_ab/libT.so timeout 200 python tools/dqn_timeline.py $b 2>&1 | tail -40; done | tee $O/timeline.txt
