#!/bin/bash
# round 5, contact g (re-used for the five-barrier form of dqn3_grad32_kernel): parity suites + same-box A / B of two library builds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_g; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_dqn3.py tests/test_gpu_dqn3w.py tests/test_gpu_bf16_tight.py tests/test_gpu_sumtree.py tests/test_gpu_run.py -q -m gpu 2>&1 | tail -8 ) > $O/tests.log 2>&1; cat $O/tests.log
L=reinforcementlearning.jl_amd/lib/librlhip.so
cp $L /tmp/lib_keep.so
for v in A B A B A B; do
    cp gpurun_ab/lib$v.so $L
    echo "lib$v: $(timeout 200 python tools/dqn3w_time.py 128 131072 2>/dev/null | tail -1)  | $(timeout 200 python tools/dqn3w_time.py 128 4096 2>/dev/null | tail -1) | $(timeout 200 python tools/dqn_fused.py 4096 3 2>/dev/null | tail -1)"
done | tee $O/ab.txt
cp /tmp/lib_keep.so $L
