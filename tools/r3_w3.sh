#!/bin/bash
# width-256 learner: parity tests of its callers + per-kernel times (rocprofv3 kernel trace of tools/ppo3w_time.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/w3; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ppo3w.py tests/test_gpu_dqn3w.py -q -x 2>&1 | tail -8 | tee $O/tests.txt
timeout 300 python tools/ppo3w_time.py 4096 128 5 256 0 2>&1 | grep -v amdgpu.ids | tee $O/time_relu.txt
timeout 300 python tools/ppo3w_time.py 4096 128 5 256 1 2>&1 | grep -v amdgpu.ids | tee $O/time_tanh.txt
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o w3 -- python $OLDPWD/tools/ppo3w_time.py 4096 128 5 256 0 > $O/prof.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -14 "$f" | cut -d, -f1-5 | tee $O/stats.txt
