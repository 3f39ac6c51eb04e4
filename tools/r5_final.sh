#!/bin/bash
# round 5 final contact: full GPU suite, unprofiled bench in the driver's form and in the default form, rocprofv3 --kernel-trace
# --stats of the default command, PMC traffic of every HBM-bound kernel of the line, VALU / MFMA instruction counters of the
# 128-wide learner tiles
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
R=$PWD; O=gpurun_out/r5_final; mkdir -p $O
rm -f gpurun_out/grad_err.jsonl gpurun_out/bench_shape_margins.jsonl
( time timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -14 ) > $O/tests.log 2>&1; cat $O/tests.log
cp gpurun_out/grad_err.jsonl gpurun_out/bench_shape_margins.jsonl $O/ 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
bash tools/prof.sh final 2>&1 | tail -10
mkdir -p $O/prof && cp gpurun_out/prof_final.log $O/ && find gpurun_out/prof_final -name "*kernel_stats.csv" -exec cp {} $O/prof/bench_kernel_stats.csv \; && find gpurun_out/prof_final -name "*kernel_trace.csv" -exec cp {} $O/prof/bench_kernel_trace.csv \;
bash tools/r5_pmc.sh > $O/pmc.log 2>&1; tail -30 $O/pmc.log
cp -r gpurun_out/r5_pmc/env.txt gpurun_out/r5_pmc/side.txt $O/ 2>/dev/null
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/$O/valu_ppo3 -o pmc -- python $R/tools/ppo3_one.py pendulum 128 3 > $R/$O/valu_ppo3.log 2>&1)
python3 tools/pmc_last.py $O/valu_ppo3 ppo3_gradT_kernel 8 2>&1 | tee $O/valu_ppo3.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/$O/valu_dqn3 -o pmc -- python $R/tools/dqn3w_time.py 128 131072 > $R/$O/valu_dqn3.log 2>&1)
python3 tools/pmc_last.py $O/valu_dqn3 dqn3_grad32_kernel 8 2>&1 | tee $O/valu_dqn3.txt
python - <<PY
import json
for f in ("bench_driver_form", "bench"):
    d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "ms_per_step", d["ms_per_step"], "value", d["value"], d.get("kernels"), "traffic", d["roofline"].get("traffic"))
PY
