#!/bin/bash
# round 4, contact I: ppo3_rollout32_kernel with actor / critic roles on separate waves -- parity + kernel time
export PYTHONPATH=$GRAFT_REPO_ROOT/reinforcementlearning.jl_amd:$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_i; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ppo3.py tests/test_gpu_learners.py -q -x -m gpu > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $GRAFT_REPO_ROOT/tools/ppo3_one.py pendulum 128 6 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); grep -i "rollout" $f | awk -F, '{print "rollout calls/total/avg:", $(NF-7), $(NF-6), $(NF-5)}'
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o p -- python $GRAFT_REPO_ROOT/tools/ppo3_one.py cartpole 128 6 > $O/prof2.log 2>&1
f=$(find $O/prof2 -name "*kernel_stats.csv" | head -1); grep -i "rollout" $f | awk -F, '{print "cartpole rollout calls/total/avg:", $(NF-7), $(NF-6), $(NF-5)}'
