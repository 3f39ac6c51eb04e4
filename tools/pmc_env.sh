#!/bin/bash
# HBM traffic of the env-step kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (kernel-trace only), mean of the
# last 20 launches of tools/envstep.py -> gpurun_out/pmc_env_<tag>.txt
tag=${1:-now}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_env_$tag.txt
: > $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_env_${tag}_$c -o pmc -- python $GRAFT_REPO_ROOT/tools/envstep.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_env_${tag}_$c.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_last.py $GRAFT_REPO_ROOT/gpurun_out/pmc_env_${tag}_$c env_step_kernel 20 >> $out
done
cat $out
