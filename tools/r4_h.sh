#!/bin/bash
# round 4, contact H: the sampling path's own Float64 log -- micro check, parity suites, kernel time, bench
export PYTHONPATH=$GRAFT_REPO_ROOT/reinforcementlearning.jl_amd:$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_h; mkdir -p $O
cd $GRAFT_REPO_ROOT
./tools/micro/log_sampling.bin
timeout 900 python -m pytest tests/test_gpu_learners.py tests/test_gpu_run.py tests/test_gpu_abi_host.py tests/test_gpu_heads.py tests/test_gpu_explorers.py tests/test_gpu_ppo3.py tests/test_gpu_parity.py -q -x -m gpu > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $GRAFT_REPO_ROOT/tools/rollout_one.py 60 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); grep -i rollout $f | awk -F, '{print "rollout calls/total/avg:", $(NF-7), $(NF-6), $(NF-5)}'
cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --no-extras > $O/bench_$i.json 2>$O/bench_$i.err; cut -c1-60,150-200 $O/bench_$i.json; done
