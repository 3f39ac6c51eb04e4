#!/bin/bash
# round 4 PMC traffic passes (FETCH_SIZE / WRITE_SIZE, SEPARATE passes, kernel-trace only) of the side kernels of
# bench.roofline_hbm_side -> gpurun_out/r4_pmc/side.txt (mean of the last 8 launches of each kernel)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4_pmc; mkdir -p $O
: > $O/side.txt
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/side_$c -o pmc -- python $R/tools/r4_pmc_side.py > $O/side_$c.log 2>&1)
  for k in "env_step_kernel<rlhip::Pendulum" "env_step_kernel<rlhip::MountainCar" adam_vec4_kernel polyak_vec4_kernel push_transition_maxpool_kernel gather_small_lane_kernel; do
    python3 tools/pmc_last.py $O/side_$c "$k" 8 >> $O/side.txt
  done
done
cat $O/side.txt
tail -2 $O/side_FETCH_SIZE.log
