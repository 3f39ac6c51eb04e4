#!/bin/bash
# round 6 contact O: dZ2 once as the FRAGMENT image (libF: the backward kernel reads it transposed from LDS with ds_read_b64_tr_b16; forward without its
# row transposition; dW2 as in round 5) against dZ2 once as rows (libE = HEAD) and round 5 (libA): parity of libF, step timings, per-kernel durations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
R=$PWD; O=gpurun_out/r6_o; mkdir -p $O
RLHIP_LIB_PATH=$PWD/gpurun_ab/libF.so timeout 900 python -m pytest tests/test_gpu_ppo3w.py tests/test_gpu_dqn3w.py tests/test_gpu_bf16_tight.py tests/test_gpu_bench_shapes_bf16.py tests/test_gpu_nstep.py -x -q -m gpu 2>&1 | tail -6 | tee $O/parity.txt
for rep in 1 2 3; do for v in A E F; do
  echo -n "$v " | tee -a $O/ab.txt
  RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so python tools/ppo3w_time.py 4096 128 5 2>&1 | grep "us per optimiser" | sed 's/.*update/update/' | tee -a $O/ab.txt
  echo -n "$v " | tee -a $O/ab.txt
  RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so python tools/dqn3w_time.py 256 131072 2>&1 | tail -1 | tee -a $O/ab.txt
done; done
for v in E F; do
  (cd /tmp && RLHIP_LIB_PATH=$R/gpurun_ab/lib$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -o p -- python $R/tools/ppo3w_time.py 4096 128 5 > $R/$O/prof_$v.log 2>&1)
  echo "== lib$v" | tee -a $O/kernels.txt
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY' | tee -a $O/kernels.txt
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].split('(')[0].replace('void rlhip::','')
    if ('ppo3w_fwd' in n or 'ppo3w_bwd' in n or 'ppo3w_dw2' in n) : print(f"{n[:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.2f}")
PY
  rm -rf $O/prof_$v
done
