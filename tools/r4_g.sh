#!/bin/bash
# round 4, contact G: the env step of categorical heads speculated by the critic wave -- parity, kernel time A / B, bench
export PYTHONPATH=$GRAFT_REPO_ROOT/reinforcementlearning.jl_amd:$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_g; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_learners.py tests/test_gpu_run.py tests/test_gpu_abi_host.py -q -x -m gpu > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
cd /tmp && export TMPDIR=/tmp
for sp in 0 1; do
  RLHIP_ROLLOUT_SPEC=$sp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sp$sp -o p -- python $GRAFT_REPO_ROOT/tools/rollout_one.py 60 > $O/prof_sp$sp.log 2>&1
  f=$(find $O/prof_sp$sp -name "*kernel_stats.csv" | head -1)
  grep -i rollout $f | awk -F, -v sp=$sp '{print "spec=" sp " rollout calls/total/avg:", $(NF-7), $(NF-6), $(NF-5)}'
done
cd $GRAFT_REPO_ROOT
for sp in 0 1 0 1; do
  RLHIP_ROLLOUT_SPEC=$sp python bench.py --no-extras > $O/bench_sp$sp.json 2>$O/bench_sp$sp.err; echo "spec=$sp $(cut -c1-60,150-230 $O/bench_sp$sp.json)"
done
