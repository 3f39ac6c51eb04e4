#!/bin/bash
# round 6 contact I: the fused write-back + draw + gather launch -- parity (sum-tree suite, config 5 at full size) and timing (bench roofline_extras)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r6_i; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_sumtree.py tests/test_gpu_config5_full.py tests/test_gpu_stackframes.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -25 ) > $O/tests.log 2>&1; cat $O/tests.log
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/small_batches.txt
import sys, json
sys.path[:0] = [".", "reinforcementlearning.jl_amd"]
import torch, rlhip, bench
out = bench.roofline_extras(torch, rlhip, hbm_only=True)
fg = out["frame_gather_u8"]
print({k: fg[k] for k in ("us_per_launch", "frac", "sample_gather_fused_us", "sample_gather_update_us")})
for k, v in fg["small_batches"].items():
    print(k, {x: v[x] for x in ("us_per_launch", "prioritized_sample_us", "priority_update_us", "sample_gather_update_us", "update_sample_gather_one_call_us")})
PY
