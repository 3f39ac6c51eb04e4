#!/bin/bash
# round 4: HBM / fabric traffic of the 256-wide learner's kernels per launch -- FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes
# (kernel-trace only); config 3 at hidden 256, 2 + 3 iterations (tools/ppo3w_time.py 4096 128 3 256)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  bash tools/pmc.sh r4w_$c "tools/ppo3w_time.py 4096 128 3 256" $c | grep -E "ppo3w|dqn3w" | cut -c1-220
done
