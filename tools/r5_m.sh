#!/bin/bash
# round 5, contact m: the 1024-thread dqn_grad_kernel + one-launch optimise!: parity subset, timeline, vec-step timing
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_m; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_learners.py tests/test_gpu_run.py tests/test_gpu_abi_host.py tests/test_gpu_bench_shapes.py tests/test_gpu_sumtree.py tests/test_gpu_edges.py -m gpu -q -k "dqn or DQN or fused or host or per or prior" 2>&1 | tail -15 | tee $O/tests.log
for b in 512 32 4096; do RLHIP_LIB_PATH=$PWD/gpurun_ab/libT.so timeout 200 python tools/dqn_timeline.py $b 2>&1 | tail -40; done | tee $O/timeline.txt
for rep in 1 2; do
  for b in 32 512 1024 2048 4096; do
    RLHIP_DQN_NO_FUSE=1 timeout 120 python tools/dqn_fused.py $b 2 2>&1 | tail -1 | sed 's/^/two-launch optimise: /'
    timeout 120 python tools/dqn_fused.py $b 2 2>&1 | tail -1 | sed 's/^/one-launch optimise: /'
  done
done | tee $O/ab.txt
