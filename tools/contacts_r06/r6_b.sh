#!/bin/bash
# round 6 contact B: fused DQN vec-step vs the oracle's agent loop; learning-curve probe; bench in the new two-form protocol
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r6_b; mkdir -p $O
rm -f gpurun_out/bench_shape_margins.jsonl
( time timeout 900 python -m pytest tests/test_gpu_dqn_agent_vs_oracle.py -x -q -m gpu 2>&1 | tail -30 ) > $O/tests.log 2>&1; cat $O/tests.log
cp gpurun_out/bench_shape_margins.jsonl $O/ 2>/dev/null; cat $O/bench_shape_margins.jsonl
timeout 900 python tools/learn_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/learn_probe.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; tail -c 400 $O/bench_driver_form.err
python - <<PY
import json
d = json.loads(open("$O/bench_driver_form.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "no_preheat", d["ms_per_step_no_preheat"], "value", d["value"], "frac", d["roofline"]["frac"])
for k in ("gather_small", "gather_small_hbm", "adam_2p22", "adam_2p26"):
    v = d["roofline_extra"][k]; print(k, v["us_per_launch"], v["frac"], v["bound"], v.get("kernel"))
PY
