#!/bin/bash
# round 5, contact o: Float64 cross-lane sums of the optimiser tails on DPP (common.h): PPO parity subset + the reduce_apply timeline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_o; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_learners.py tests/test_gpu_run.py tests/test_gpu_bench_shapes.py -m gpu -q 2>&1 | tail -8 | tee $O/tests.log
RLHIP_TICK_NS=0.415 timeout 300 python tools/grad_timeline.py 2>&1 | tail -22 | tee $O/grad_timeline.txt
