#!/bin/bash
# round 6 contact L: per-kernel durations (rocprofv3 --kernel-trace --stats) of the 256-wide PPO step, round-5 kernel (libA) vs merged barriers (libC)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
R=$PWD; O=gpurun_out/r6_l; mkdir -p $O
for v in A C A C; do
  (cd /tmp && RLHIP_LIB_PATH=$R/gpurun_ab/lib$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -o p -- python $R/tools/ppo3w_time.py 4096 128 5 > $R/$O/prof_$v.log 2>&1)
  echo "== lib$v" | tee -a $O/kernels.txt
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY' | tee -a $O/kernels.txt
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].split('(')[0].replace('void rlhip::','')
    if 'ppo3w' in n and 'rollout' not in n: print(f"{n[:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.2f}")
PY
  rm -rf $O/prof_$v
done
