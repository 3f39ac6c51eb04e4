#!/bin/bash
# round 6 contact Q: libF (fragment image, unpadded LDS copy, 89 KB) / libG (padded copy, 95 KB) / libH (unpadded copy in a 95 KB allocation): step
# timings in permuted orders, per-kernel durations in both orders
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
R=$PWD; O=gpurun_out/r6_q; mkdir -p $O
for order in "F G H" "H G F" "G F H" "F H G" "G H F" "H F G"; do for v in $order; do
  echo -n "$v " | tee -a $O/ab.txt
  RLHIP_LIB_PATH=$PWD/gpurun_ab/lib$v.so python tools/ppo3w_time.py 4096 128 5 2>&1 | grep "us per optimiser" | sed 's/.*update/update/' | tee -a $O/ab.txt
done; done
for v in G F H G F; do
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
