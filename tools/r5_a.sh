#!/bin/bash
# round 5, first GPU contact: full GPU suite on the record ring / ABI 2, the default bench line, Adam chunks-per-thread A / B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_a; mkdir -p $O
rm -f gpurun_out/grad_err.jsonl gpurun_out/bench_shape_margins.jsonl
( time timeout 1700 python -m pytest tests -q -m gpu -x 2>&1 | tail -40 ) > $O/tests.log 2>&1; tail -25 $O/tests.log
cp gpurun_out/grad_err.jsonl gpurun_out/bench_shape_margins.jsonl $O/ 2>/dev/null
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"].get("traffic"))
x = d.get("roofline_extra", {})
for k in ("gather_small", "adam_2p22", "adam_2p26"):
    for kk, v in x.items():
        if k in kk: print(kk, v.get("us_per_launch"), v.get("frac"))
print(json.dumps(x.get("dqn_cartpole_4096env", {}), indent=None)[:900])
PY
L=reinforcementlearning.jl_amd/lib/librlhip.so
cp $L /tmp/lib_keep.so
for v in A B A B; do
    cp gpurun_ab/lib$v.so $L
    echo "adam chunks lib$v: $(timeout 300 python tools/adam_grid_ab.py 2>/dev/null | tail -1)"
done
cp /tmp/lib_keep.so $L
