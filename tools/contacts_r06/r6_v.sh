#!/bin/bash
# round 6 contact V: the forward kernel's fragment stores as 16-byte stores behind v_permlane32_swap (libK) against 8-byte stores (libF = HEAD~): parity, bursts, per-kernel durations, sustained
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
R=$PWD; O=gpurun_out/r6_v; mkdir -p $O; rm -f $O/*
for v in K; do
RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so timeout 900 python -m pytest tests/test_gpu_ppo3w.py tests/test_gpu_dqn3w.py tests/test_gpu_bf16_tight.py tests/test_gpu_bench_shapes_bf16.py -x -q -m gpu 2>&1 | tail -4 | tee -a $O/parity.txt
done
for order in "F K" "K F" "F K"; do for v in $order; do
  echo -n "$v " | tee -a $O/ab.txt
  RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so python tools/ppo3w_time.py 4096 128 5 2>&1 | grep "us per optimiser" | sed 's/.*update/update/' | tee -a $O/ab.txt
  echo -n "$v " | tee -a $O/ab.txt
  RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so python tools/dqn3w_time.py 256 131072 2>&1 | tail -1 | tee -a $O/ab.txt
done; done
for v in F K; do
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
for v in F K F K; do
  RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so timeout 300 python tools/power_probe.py 4 ppo3w > $O/probe_ppo3w_$v.txt 2>&1
  grep -E "STEADY" $O/probe_ppo3w_$v.txt | tee -a $O/steady.txt
done
