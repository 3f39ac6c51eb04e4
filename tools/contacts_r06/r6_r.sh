#!/bin/bash
# round 6 contact R: power / clock of the chip while the 256-wide optimiser step runs back to back for 4 s (libF and libG), and idle
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r6_r; mkdir -p $O
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40 > $O/smi_idle.txt
for v in F G; do
  RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so timeout 300 python tools/power_probe.py 4 ppo3w > $O/probe_$v.txt 2>&1
done
( RLHIP_LIB_PATH=$PWD/gpurun_ab/libF.so timeout 300 python tools/power_probe.py 6 ppo3w > $O/probe_F2.txt 2>&1 & sleep 22; for i in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" ; sleep 1; done > $O/smi_busy.txt; wait )
tail -70 $O/probe_F.txt; tail -50 $O/probe_G.txt; cat $O/smi_idle.txt $O/smi_busy.txt
