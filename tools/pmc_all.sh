#!/bin/bash
# PMC traffic passes (FETCH_SIZE / WRITE_SIZE in SEPARATE rocprofv3 passes, kernel-trace only) of every HBM-bound kernel in
# the bench line; prints the PMC_TRAFFIC / PMC_SIDE entries for bench.py (2 x FETCH + WRITE, mean of the last launches)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
R=$PWD; O=gpurun_out/pmc_all; mkdir -p $O
bash tools/pmc_env.sh final 2>&1 | tail -3
cp gpurun_out/pmc_env_final.txt $O/env.txt
: > $O/side.txt
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/side_$c -o pmc -- python $R/tools/pmc_all.py > $R/$O/side_$c.log 2>&1)
  for k in gae_vec4_kernel "env_step_kernel<rlhip::Pendulum" "env_step_kernel<rlhip::MountainCar" push_transition_maxpool_kernel; do
    python3 tools/pmc_last.py $O/side_$c "$k" 8 >> $O/side.txt
  done
  # the record gather runs at two operating points with the same grid: the last launches of pmc_all.py are the 4.3 GB ring
  # (gather_small_hbm); the Infinity-Cache-resident ring (gather_small) gets its own pass below
  echo "gather_small_hbm:" >> $O/side.txt
  python3 tools/pmc_last.py $O/side_$c gather_rec_kernel 8 >> $O/side.txt
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/gs_$c -o pmc -- python $R/tools/gather_rec_time.py > $R/$O/gs_$c.log 2>&1)
  echo "gather_small:" >> $O/side.txt
  python3 tools/pmc_last.py $O/gs_$c gather_rec_kernel 8 >> $O/side.txt
  # kernels launched at several sizes: read them at the size of their bench entry (Grid_Size in threads)
  python3 tools/pmc_last.py $O/side_$c gather_frames_kernel 8 $((4096 * 256)) >> $O/side.txt    # batch 4096: one workgroup per sample
  python3 tools/pmc_last.py $O/side_$c gather_stacked_kernel 8 $((4096 * 256)) >> $O/side.txt
  python3 tools/pmc_last.py $O/side_$c adam_vec4_kernel 8 $((1 << 24)) >> $O/side.txt           # 2^26 parameters: one 16-byte chunk per thread
  python3 tools/pmc_last.py $O/side_$c adam_vec4_kernel 8 $((1 << 20)) >> $O/side.txt           # 2^22
  python3 tools/pmc_last.py $O/side_$c polyak_vec4_kernel 8 $((1 << 24)) >> $O/side.txt
  python3 tools/pmc_last.py $O/side_$c polyak_vec4_kernel 8 $((1 << 20)) >> $O/side.txt
done
cat $O/side.txt; tail -2 $O/side_FETCH_SIZE.log
