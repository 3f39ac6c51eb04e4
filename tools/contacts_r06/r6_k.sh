#!/bin/bash
# round 6 contact K: stamped timelines of the 256-wide forward kernel (actor), shipped schedule vs the merged-barrier schedule
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r6_k; mkdir -p $O
for rep in 1 2; do
echo "shipped schedule (3 barriers per tile):" | tee -a $O/timeline.txt
RLHIP_LIB_PATH=$PWD/gpurun_ab/libT.so python tools/w3timing_fwd.py 2>&1 | grep -v amdgpu | tee -a $O/timeline.txt
echo "merged-barrier schedule (2 barriers per tile):" | tee -a $O/timeline.txt
W3_MERGED=1 RLHIP_LIB_PATH=$PWD/gpurun_ab/libTm.so python tools/w3timing_fwd.py 2>&1 | grep -v amdgpu | tee -a $O/timeline.txt
done
