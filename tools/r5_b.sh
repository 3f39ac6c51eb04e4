#!/bin/bash
# round 5, second GPU contact: full GPU suite (no -x), gather variants, Adam chunk A / B, PMC traffic of env-step + side kernels, bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
R=$PWD; O=gpurun_out/r5_b; mkdir -p $O
rm -f gpurun_out/grad_err.jsonl gpurun_out/bench_shape_margins.jsonl
( time timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -60 ) > $O/tests.log 2>&1; tail -30 $O/tests.log
cp gpurun_out/grad_err.jsonl gpurun_out/bench_shape_margins.jsonl $O/ 2>/dev/null
for v in 0 1 2 3 0 1 2 3; do RLHIP_GATHER_VARIANT=$v timeout 120 python tools/r5_gather.py 2>/dev/null | tail -1; done | tee $O/gather_variants.txt
R5_GATHER_CAP=16384 timeout 120 python tools/r5_gather.py 2>/dev/null | tail -1 | tee -a $O/gather_variants.txt
L=reinforcementlearning.jl_amd/lib/librlhip.so
cp $L /tmp/lib_keep.so
for v in A B A B; do
    cp gpurun_ab/lib$v.so $L
    echo "adam chunks lib$v: $(timeout 300 python tools/adam_grid_ab.py 2>/dev/null | tail -1)"
done | tee $O/adam_ab.txt
cp /tmp/lib_keep.so $L
# PMC traffic: env-step (the bench's roofline kernel) and the side kernels, FETCH_SIZE / WRITE_SIZE in separate passes
bash tools/pmc_env.sh r5 2>&1 | tail -4
: > $O/side.txt
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/side_$c -o pmc -- python $R/tools/r4_pmc_side.py > $R/$O/side_$c.log 2>&1)
  for k in "env_step_kernel<rlhip::Pendulum" "env_step_kernel<rlhip::MountainCar" adam_vec4_kernel polyak_vec4_kernel push_transition_maxpool_kernel gather_rec_kernel; do
    python3 tools/pmc_last.py $O/side_$c "$k" 8 >> $O/side.txt
  done
done
cat $O/side.txt
cp gpurun_out/pmc_env_r5.txt $O/ 2>/dev/null
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["frac"], d["roofline"].get("traffic"), d.get("kernels"))
x = d.get("roofline_extra", {})
for kk, v in x.items():
    if isinstance(v, dict) and "frac" in v: print(kk, v.get("us_per_launch"), v.get("frac"))
PY
