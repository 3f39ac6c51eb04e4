#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/pmc_env.sh r3 2>&1 | tail -4
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_side_$c -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_side.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_side_$c.log 2>&1)
  for k in gae_vec4_kernel gather_frames_kernel gather_stacked_kernel; do python3 tools/pmc_last.py gpurun_out/pmc_side_$c $k 8; done
done | tee gpurun_out/pmc_side_r3.txt
