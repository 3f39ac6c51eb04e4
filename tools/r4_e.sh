#!/bin/bash
# round 4, contact E: two-wave rollout as the only wide path -- parity (learner / run / abi-host suites), kernel time, bench in the driver's form
export PYTHONPATH=$GRAFT_REPO_ROOT/reinforcementlearning.jl_amd:$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_e; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_learners.py tests/test_gpu_run.py tests/test_gpu_abi_host.py -q -x -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $GRAFT_REPO_ROOT/tools/rollout_one.py 60 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); grep -i rollout $f | awk -F, '{print "rollout calls/total/avg:", $(NF-7), $(NF-6), $(NF-5)}'
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2>$O/bench_20.err; cut -c1-230 $O/bench_20.json
python bench.py --no-extras > $O/bench_200.json 2>$O/bench_200.err; cut -c1-230 $O/bench_200.json
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4_e/"
d=json.loads(open(O+"bench_20.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["kernels"], d["roofline"]["frac"], d.get("legs_order"))
PY
