#!/bin/bash
# round 4, contact K: LDS-only barrier in the two-wave rollout -- parity, kernel time, same-box A / B of the iteration
export PYTHONPATH=$GRAFT_REPO_ROOT/reinforcementlearning.jl_amd:$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_k; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_learners.py tests/test_gpu_run.py tests/test_gpu_abi_host.py -q -x -m gpu > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $GRAFT_REPO_ROOT/tools/rollout_one.py 60 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); grep -i rollout $f | awk -F, '{print "rollout calls/total/avg:", $(NF-7), $(NF-6), $(NF-5)}'
cd $GRAFT_REPO_ROOT
bash tools/ab.sh
