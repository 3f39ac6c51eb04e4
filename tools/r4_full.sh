#!/bin/bash
# round 4: full GPU suite + default bench line (+ optional profile)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
tag=${1:-full}
O=gpurun_out/r4_$tag; mkdir -p $O
rm -f gpurun_out/grad_err.jsonl
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -15 $O/tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","updates_per_sec")}, d["kernels"])
print("roofline", {k:d["roofline"][k] for k in ("us_per_launch","achieved","frac")})
for k,v in d["roofline_extra"].items():
    print(k, {kk:v[kk] for kk in ("us_per_launch","frac","achieved","per_microbatch_us","ms_per_iteration","frac_of_bf16_peak","sample_gather_update_us","small_batches","ms_per_vec_step") if kk in v})
PY
cp gpurun_out/grad_err.jsonl $O/ 2>/dev/null
