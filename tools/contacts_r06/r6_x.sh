#!/bin/bash
# round 6 contact X: the dW2 kernel's pipelined pass again (H1^T of tile t + 1 built under the MFMAs of tile t), with the instruction order PINNED by
# scheduling barriers: f32 MFMAs of layer 1 first, the activation / pack / ds_write behind k-step 1 (libM) or 2 (libN) of the bf16 MFMAs -- against libF (HEAD)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp RLHIP_W3_DZF_PAD=0
R=$PWD; O=gpurun_out/r6_x; mkdir -p $O; rm -f $O/*
for v in M N; do RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so timeout 900 python -m pytest tests/test_gpu_ppo3w.py tests/test_gpu_dqn3w.py -x -q -m gpu 2>&1 | tail -1 | tee -a $O/parity.txt; done
for v in F M N F M N; do
  (cd /tmp && RLHIP_LIB_PATH=$R/gpurun_ab/lib$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -o p -- python $R/tools/ppo3w_time.py 4096 128 5 > $R/$O/prof_$v.log 2>&1)
  echo "== lib$v $(grep 'us per optimiser' $O/prof_$v.log | sed 's/.*update/update/' | cut -c1-60)" | tee -a $O/kernels.txt; python3 tools/kstats.py $O/prof_$v dw2 | tee -a $O/kernels.txt; rm -rf $O/prof_$v
done
