#!/bin/bash
# round 5, contact h: the prioritized draw by a whole wavefront (four tree levels per round trip) -- ring / sum-tree suites, the
# bench's small-batch numbers, then the PMC traffic passes again (csrc/ring.hip changed: the sha guard of PMC_SIDE)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_h; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_sumtree.py tests/test_gpu_stackframes.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_run.py -q -m gpu 2>&1 | tail -8 ) > $O/tests.log 2>&1; cat $O/tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], d.get("kernels"))
x = d["roofline_extra"]
fg = x["frame_gather_u8"]
print({k: fg[k] for k in ("us_per_launch", "frac", "prioritized_sample_us", "priority_update_us", "sample_gather_fused_us", "sample_gather_update_us", "traffic")})
print(fg["small_batches"])
PY
bash tools/r5_pmc.sh > $O/pmc.log 2>&1; tail -26 $O/pmc.log
cp gpurun_out/r5_pmc/env.txt gpurun_out/r5_pmc/side.txt $O/ 2>/dev/null
