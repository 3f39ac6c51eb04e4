#!/bin/bash
# the round's final contact: full GPU suite, unprofiled bench in the driver's form and in the default form, rocprofv3 --kernel-trace
# --stats of the default command, PMC traffic of every HBM-bound kernel of the line (tools/pmc_all.sh)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
R=$PWD; O=gpurun_out/final; mkdir -p $O
rm -f gpurun_out/grad_err.jsonl gpurun_out/bench_shape_margins.jsonl
if [ "$1" != "nosuite" ]; then
( time timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -14 ) > $O/tests.log 2>&1; cat $O/tests.log
cp gpurun_out/grad_err.jsonl gpurun_out/bench_shape_margins.jsonl $O/ 2>/dev/null
fi
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
bash tools/prof.sh final 2>&1 | tail -10
mkdir -p $O/prof && cp gpurun_out/prof_final.log $O/ && find gpurun_out/prof_final -name "*kernel_stats.csv" -exec cp {} $O/prof/bench_kernel_stats.csv \; && find gpurun_out/prof_final -name "*kernel_trace.csv" -exec cp {} $O/prof/bench_kernel_trace.csv \;
bash tools/pmc_all.sh > $O/pmc.log 2>&1; tail -40 $O/pmc.log
cp -r gpurun_out/pmc_all/env.txt gpurun_out/pmc_all/side.txt $O/ 2>/dev/null
# fabric traffic per kernel of the 256-wide PPO optimiser step at HEAD (FETCH_SIZE x 2 / WRITE_SIZE in separate passes; units: 32 B / 64 B per pmc_last.py)
rm -f $O/ppo3w_traffic.txt
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_w_$c -o pmc -- python $R/tools/ppo3w_time.py 4096 128 5 > $R/$O/pmc_w_$c.log 2>&1)
  for k in ppo3w_gather_rec_kernel ppo3w_fwd_kernel ppo3w_bwd_kernel ppo3w_dw2_kernel ppo3w_reduce_sumsq_kernel ppo3w_adam_pack_kernel; do python3 tools/pmc_last.py $O/pmc_w_$c $k 32 | tee -a $O/ppo3w_traffic.txt; done
  rm -rf $O/pmc_w_$c
done
python - <<PY
import json
for f in ("bench_driver_form", "bench"):
    d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "ms_per_step", d["ms_per_step"], "no_preheat", d["ms_per_step_no_preheat"], "value", d["value"], d.get("kernels"), "traffic", d["roofline"].get("traffic"))
    print("   sustained_clock", json.dumps(d.get("sustained_clock")))
    for k, v in d.get("roofline_extra", {}).items():
        if isinstance(v, dict) and "frac" in v: print("   ", k, v.get("us_per_launch"), v["frac"], v.get("traffic_ratio"))
PY
