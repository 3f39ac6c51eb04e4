#!/bin/bash
# (historical: the RLHIP_* A / B hook this script toggles was removed once the result was in profiles/raw_r05/)
# round 5, sixth contact: small-batch prioritized path (one-wave sum-tree update, LDS-cached descent) -- sum-tree suites + the bench's small-batch numbers
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_f; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_sumtree.py tests/test_gpu_stackframes.py tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -20 ) > $O/tests.log 2>&1; tail -8 $O/tests.log
for v in 0 1 0 1; do echo "no_small=$v $(RLHIP_SUMTREE_NO_SMALL=$v timeout 200 python tools/sumtree_update_time.py 2>/dev/null | tail -3 | tr '\n' ' ')"; done | tee $O/update_ab.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], d.get("kernels"))
x = d["roofline_extra"]
print(json.dumps(x["frame_gather_u8"], indent=None)[:2500])
PY
