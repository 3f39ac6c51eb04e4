#!/bin/bash
# round 6 contact AA: the forward kernel's A fragments software-pipelined one pair of k-steps ahead (register ring + scheduling barriers) = libQ against libF (HEAD):
# parity of libQ, step / gradient timings, per-kernel durations, sustained
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
R=$PWD; O=gpurun_out/r6_aa; mkdir -p $O; rm -f $O/*
RLHIP_LIB_PATH=$PWD/gpurun_ab/libQ.so timeout 1200 python -m pytest tests/test_gpu_ppo3w.py tests/test_gpu_dqn3w.py tests/test_gpu_bf16_tight.py tests/test_gpu_bench_shapes_bf16.py -x -q -m gpu 2>&1 | tail -2 | tee -a $O/parity.txt
for order in "F Q" "Q F" "F Q"; do for v in $order; do
  echo -n "$v " | tee -a $O/ab.txt
  RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so python tools/dqn3w_time.py 256 131072 2>&1 | tail -1 | tee -a $O/ab.txt
  echo -n "$v " | tee -a $O/ab.txt
  RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so python tools/ppo3w_time.py 4096 128 5 2>&1 | grep "us per optimiser" | sed 's/.*T 128: //' | tee -a $O/ab.txt
done; done
for v in F Q; do
  (cd /tmp && RLHIP_LIB_PATH=$R/gpurun_ab/lib$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -o p -- python $R/tools/ppo3w_time.py 4096 128 5 > $R/$O/prof_$v.log 2>&1)
  echo "== lib$v ppo3w" | tee -a $O/kernels.txt; python3 tools/kstats.py $O/prof_$v fwd | tee -a $O/kernels.txt; rm -rf $O/prof_$v
  (cd /tmp && RLHIP_LIB_PATH=$R/gpurun_ab/lib$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -o p -- python $R/tools/dqn3w_time.py 256 131072 > $R/$O/prof_$v.log 2>&1)
  echo "== lib$v dqn3w" | tee -a $O/kernels.txt; python3 tools/kstats.py $O/prof_$v fwd | tee -a $O/kernels.txt; rm -rf $O/prof_$v
done
for v in F Q F Q; do
  RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so timeout 300 python tools/power_probe.py 3 ppo3w 2>&1 | grep STEADY | tee -a $O/steady.txt
done
