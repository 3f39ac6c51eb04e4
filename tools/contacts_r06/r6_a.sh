#!/bin/bash
# round 6 contact A: the new bench-shape parity tests (config 5 at full size, the four bf16 learners at 131072 samples) + baseline bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r6_a; mkdir -p $O
rm -f gpurun_out/grad_err.jsonl gpurun_out/bench_shape_margins.jsonl
nproc > $O/nproc.txt; free -g >> $O/nproc.txt
( time timeout 1500 python -m pytest tests/test_gpu_bench_shapes_bf16.py tests/test_gpu_config5_full.py -x -q -m gpu --durations=15 2>&1 | tail -40 ) > $O/tests.log 2>&1; cat $O/tests.log
cp gpurun_out/grad_err.jsonl gpurun_out/bench_shape_margins.jsonl $O/ 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; tail -c 400 $O/bench_driver_form.err
python - <<PY
import json
d = json.loads(open("$O/bench_driver_form.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"])
for k, v in d.get("roofline_extra", {}).items():
    if isinstance(v, dict) and "frac" in v: print(k, v.get("us_per_launch"), v["frac"])
PY
