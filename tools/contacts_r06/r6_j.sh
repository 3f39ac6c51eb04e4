#!/bin/bash
# round 6 contact J: the 256-wide forward kernel with the MERGED barrier schedule (libC) against the restructured default (libB) and the
# round-5 kernel (libA): parity of libC, then same-box timings of the PPO step and the DQN gradient, three alternations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r6_j; mkdir -p $O
RLHIP_LIB_PATH=$PWD/gpurun_ab/libC.so timeout 900 python -m pytest tests/test_gpu_ppo3w.py tests/test_gpu_dqn3w.py tests/test_gpu_bf16_tight.py -x -q -m gpu 2>&1 | tail -4 | tee $O/parity_merged.txt
for rep in 1 2 3; do for v in A B C; do
  echo -n "$v " | tee -a $O/ab.txt
  RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so python tools/ppo3w_time.py 4096 128 5 2>&1 | grep "us per optimiser" | sed 's/.*update/update/' | tee -a $O/ab.txt
  echo -n "$v " | tee -a $O/ab.txt
  RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so python tools/dqn3w_time.py 256 131072 2>&1 | tail -1 | tee -a $O/ab.txt
done; done
