#!/bin/bash
# round 5, mid-session check: full GPU suite + the bench line in the driver's form and in the default form
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_mid; mkdir -p $O
( time timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -14 ) > $O/tests.log 2>&1; cat $O/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<PY
import json
for f in ("bench_driver_form", "bench"):
    d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "ms_per_step", d["ms_per_step"], "value", d["value"], d.get("kernels"), "traffic", d["roofline"].get("traffic"), d["roofline"].get("frac"))
    x = d["roofline_extra"]
    for k in ("dqn_vec_step", "dqn3_grad_mfma", "ppo3_grad_mfma", "gather_small"):
        if k in x: print("  ", k, json.dumps(x[k])[:600])
PY
