#!/bin/bash
# round 5, fourth contact: full GPU suite on the 64-byte record ring, gather timing, default bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
R=$PWD; O=gpurun_out/r5_d; mkdir -p $O
rm -f gpurun_out/grad_err.jsonl gpurun_out/bench_shape_margins.jsonl
( time timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -60 ) > $O/tests.log 2>&1; tail -30 $O/tests.log
cp gpurun_out/grad_err.jsonl gpurun_out/bench_shape_margins.jsonl $O/ 2>/dev/null
for i in 1 2; do timeout 120 python tools/r5_gather.py 2>/dev/null | tail -1; done | tee $O/gather.txt
R5_GATHER_CAP=16384 timeout 120 python tools/r5_gather.py 2>/dev/null | tail -1 | tee -a $O/gather.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"].get("traffic"), d.get("kernels"))
x = d.get("roofline_extra", {})
for kk, v in x.items():
    if isinstance(v, dict) and "frac" in v: print(kk, v.get("us_per_launch"), v.get("frac"), v.get("traffic_ratio"))
print(json.dumps(x["dqn_cartpole_4096env"])[:1800])
PY
