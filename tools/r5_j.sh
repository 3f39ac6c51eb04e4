#!/bin/bash
# round 5, contact j: stamped timeline of a steady-state tile of dqn3_grad32_kernel (timing build gpurun_ab/libT.so)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_j; mkdir -p $O
for b in 131072 131072 4096; do RLHIP_LIB_PATH=$PWD/gpurun_ab/libT.so timeout 200 python tools/d3g32_timeline.py $b 2>&1 | tail -22; done | tee $O/timeline.txt
