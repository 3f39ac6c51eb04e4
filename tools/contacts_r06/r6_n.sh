#!/bin/bash
# round 6 contact N: dZ2 once with the fragment gather software-pipelined (libE = in-tree default) vs the first shipped form (libD) vs round 5 (libA):
# parity (libE), per-kernel durations, step timings; then fabric traffic per kernel of the step at HEAD (FETCH_SIZE / WRITE_SIZE, separate passes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
R=$PWD; O=gpurun_out/r6_n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ppo3w.py tests/test_gpu_dqn3w.py tests/test_gpu_bf16_tight.py tests/test_gpu_bench_shapes_bf16.py tests/test_gpu_nstep.py -x -q -m gpu 2>&1 | tail -3 | tee $O/parity.txt
for rep in 1 2 3; do for v in A D E; do
  echo -n "$v " | tee -a $O/ab.txt
  RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so python tools/ppo3w_time.py 4096 128 5 2>&1 | grep "us per optimiser" | sed 's/.*update/update/' | tee -a $O/ab.txt
  echo -n "$v " | tee -a $O/ab.txt
  RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so python tools/dqn3w_time.py 256 131072 2>&1 | tail -1 | tee -a $O/ab.txt
done; done
for v in D E; do
  (cd /tmp && RLHIP_LIB_PATH=$R/gpurun_ab/lib$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -o p -- python $R/tools/ppo3w_time.py 4096 128 5 > $R/$O/prof_$v.log 2>&1)
  echo "== lib$v" | tee -a $O/kernels.txt
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY' | tee -a $O/kernels.txt
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].split('(')[0].replace('void rlhip::','')
    if 'ppo3w' in n and 'rollout' not in n and 'rec' not in n and 'pack_kernel' != n[-11:]: print(f"{n[:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.2f}")
PY
  rm -rf $O/prof_$v
done
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_$c -o pmc -- python $R/tools/ppo3w_time.py 4096 128 5 > $R/$O/pmc_$c.log 2>&1)
  for k in ppo3w_gather_rec_kernel ppo3w_fwd_kernel ppo3w_bwd_kernel ppo3w_dw2_kernel ppo3w_reduce_sumsq_kernel ppo3w_adam_pack_kernel; do python3 tools/pmc_last.py $O/pmc_$c $k 32 | tee -a $O/traffic.txt; done
  rm -rf $O/pmc_$c
done
