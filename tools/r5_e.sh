#!/bin/bash
# (historical: the RLHIP_* A / B hook this script toggles was removed once the result was in profiles/raw_r05/)
# round 5, fifth contact: the update call without its pack launch -- learner parity suites + same-box A / B of the headline step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_e; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_learners.py tests/test_gpu_bench_shapes.py tests/test_gpu_run.py tests/test_gpu_abi_host.py tests/test_gpu_micro.py -q -m gpu 2>&1 | tail -30 ) > $O/tests.log 2>&1; tail -12 $O/tests.log
for v in 0 1 0 1 0 1; do
    r=$(RLHIP_PPO_PACK_LAUNCH=$v timeout 300 python bench.py --steps 300 --warmup 30 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('final_loss'))")
    echo "pack_launch=$v $r"
done | tee $O/ab.txt
python smoke_run.py 2>/dev/null; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
