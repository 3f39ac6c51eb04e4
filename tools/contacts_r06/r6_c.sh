#!/bin/bash
# round 6 contact C: full GPU suite after the ADVICE fixes (ring push protocol, optimiser slots, load_once) + the new agent-loop / learn tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r6_c; mkdir -p $O
rm -f gpurun_out/grad_err.jsonl gpurun_out/bench_shape_margins.jsonl
( time timeout 2400 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -60 ) > $O/tests.log 2>&1; cat $O/tests.log
cp gpurun_out/grad_err.jsonl gpurun_out/bench_shape_margins.jsonl $O/ 2>/dev/null
grep -h "DQN\|PPO learns\|fused DQN" $O/bench_shape_margins.jsonl | cut -c1-700
