#!/bin/bash
# round 6 contact S: power / clock while each workload runs back to back for 4 s: the headline step, the 256-wide DQN gradient, the HBM-bound env step,
# the 256-wide optimiser step with libE (dZ2 as rows) / libF / libG
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r6_s; mkdir -p $O; rm -f $O/*
for m in headline dqn3w envstep; do
  RLHIP_LIB_PATH=$PWD/gpurun_ab/libF.so timeout 300 python tools/power_probe.py 4 $m > $O/probe_$m.txt 2>&1
  grep -E "matched|STEADY" $O/probe_$m.txt
done
for v in E G F E G F; do
  RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so timeout 300 python tools/power_probe.py 4 ppo3w > $O/probe_ppo3w_$v.txt 2>&1
  grep -E "STEADY" $O/probe_ppo3w_$v.txt
done
