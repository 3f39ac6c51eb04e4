#!/bin/bash
# two-layer PPO tile with the f32-MFMA phase 1a: micro checks, parity suites of its callers, bench, timeline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/tile; mkdir -p $O
timeout 120 tools/micro/tanh_sel.bin 2>&1 | tee $O/tanh.txt
timeout 900 python -m pytest tests/test_gpu_learners.py -x -q 2>&1 | tail -8 | tee $O/tests.txt
timeout 900 python -m pytest tests/test_gpu_persist.py -x -q 2>&1 | tail -8 | tee -a $O/tests.txt
timeout 300 python bench.py --steps 200 --warmup 20 --no-extras > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
RLHIP_PPO_PERSIST=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-extras > $O/bench_persist.json 2> $O/bench_persist.err; cut -c1-300 $O/bench_persist.json
RLHIP_PPO_PERSIST=1 timeout 300 python tools/persist_timeline.py 2>&1 | grep -v "workgroup 0:\|amdgpu.ids" | tee $O/timeline.txt
timeout 300 python tools/grad_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/grad_timeline.txt
