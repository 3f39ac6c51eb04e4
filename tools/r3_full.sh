#!/bin/bash
# full GPU suite + default bench line (+ optional rocprof of the bench)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/${1:-full}
mkdir -p $O
rm -f gpurun_out/grad_err.jsonl
( time timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) > $O/tests.log 2>&1
cat $O/tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"])
print("kernels", d.get("kernels"))
print("roofline", {k: d["roofline"][k] for k in ("us_per_launch", "achieved", "frac", "traffic")})
for k, v in d.get("roofline_extra", {}).items():
    if isinstance(v, dict):
        print(k, {kk: v[kk] for kk in v if kk in ("us_per_launch", "achieved", "frac", "per_microbatch_us", "update_us", "rollout_us", "ms_per_iteration", "us_per_vec_step", "env_steps_per_sec")})
print("cpu", d.get("cpu_baseline"))
PY
