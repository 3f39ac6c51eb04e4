#!/bin/bash
# round 4 final contact: full GPU suite, unprofiled default bench, rocprofv3 --kernel-trace --stats of the same command
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r4_final; mkdir -p $O
rm -f gpurun_out/grad_err.jsonl
( time timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 ) > $O/tests.log 2>&1; cat $O/tests.log
cp gpurun_out/grad_err.jsonl $O/ 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
bash tools/prof.sh final 2>&1 | tail -10
mkdir -p $O/prof && cp gpurun_out/prof_final.log $O/ && find gpurun_out/prof_final -name "*kernel_stats.csv" -exec cp {} $O/prof/bench_kernel_stats.csv \; && find gpurun_out/prof_final -name "*kernel_trace.csv" -exec cp {} $O/prof/bench_kernel_trace.csv \;
python - <<PY
import json
for f in ("bench_driver_form", "bench"):
    d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "ms_per_step", d["ms_per_step"], "value", d["value"], d.get("kernels"))
PY
