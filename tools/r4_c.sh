#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4_c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sumtree.py tests/test_gpu_parity.py tests/test_gpu_stackframes.py -q -x > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -6 $O/tests.log
python tools/r4_side.py 2>/dev/null | tail -1
python - <<'PY'
import os, sys, json
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"reinforcementlearning.jl_amd")]
os.environ["RLHIP_BENCH_RING_SLOTS"]=str(1<<20)
import torch, rlhip, bench
# only the config-5 part of roofline_extras is wanted: run it and print that entry
out = bench.roofline_extras(torch, rlhip)
print(json.dumps(out["frame_gather_u8"]))
PY
