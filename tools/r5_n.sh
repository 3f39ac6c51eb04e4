#!/bin/bash
# round 5, contact n: full GPU suite on the DPP-row dqn_grad_kernel + one-launch optimise!, timeline, vec-step timing, C host
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_n; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/tests.log
for b in 512 32 2048; do RLHIP_LIB_PATH=$PWD/gpurun_ab/libT.so timeout 200 python tools/dqn_timeline.py $b 2>&1 | tail -40; done | tee $O/timeline.txt
for rep in 1 2; do
  for b in 32 512 2048 4096; do
    RLHIP_DQN_NO_FUSE=1 timeout 120 python tools/dqn_fused.py $b 2 2>&1 | tail -1 | sed 's/^/two-launch optimise: /'
    timeout 120 python tools/dqn_fused.py $b 2 2>&1 | tail -1 | sed 's/^/one-launch optimise: /'
  done
done | tee $O/ab.txt
cat gpurun_out/abi_host_time.json | tee $O/abi_host_time.json
