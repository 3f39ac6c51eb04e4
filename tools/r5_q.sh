#!/bin/bash
# round 5, contact q: the eight-wave DQN learner tile (dqn3_grad16_kernel, RLHIP_DQN3_G16=1): parity + timing against dqn3_grad32_kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_q; mkdir -p $O
tools/micro/mfma_bf16_16x16x32.bin | tee $O/micro.txt
RLHIP_DQN3_G16=1 timeout 1500 python -m pytest tests/test_gpu_dqn3.py -m gpu -q -x 2>&1 | tail -25 | tee $O/tests.log
for rep in 1 2; do
  for b in 131072 32768 16384 4096 512; do
    echo "grad32/128: $(timeout 120 python tools/dqn3w_time.py 128 $b 2>&1 | tail -1)"
    echo "grad16:     $(RLHIP_DQN3_G16=1 timeout 120 python tools/dqn3w_time.py 128 $b 2>&1 | tail -1)"
  done
  echo "grad32: $(timeout 120 python tools/dqn_fused.py 512 3 2>&1 | tail -1)"
  echo "grad16: $(RLHIP_DQN3_G16=1 timeout 120 python tools/dqn_fused.py 512 3 2>&1 | tail -1)"
done | tee $O/ab.txt
