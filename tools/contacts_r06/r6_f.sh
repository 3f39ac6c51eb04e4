#!/bin/bash
# round 6 contact F: fabric-side counters of the stream-mix kernels of tools/micro/adam_stream (Adam 1R+3RW vs Polyak 1R+1RW vs 3R+1RW ...)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
R=$PWD; O=gpurun_out/r6_f; mkdir -p $O
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' | head -c 6000) > $O/tcc_counters.txt; wc -c $O/tcc_counters.txt
for set in "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_EA0_RDREQ" "TCC_TAG_STALL TCC_BUSY TCC_EA0_WRREQ_64B" "TCC_EA0_WR_UNCACHED_32B TCC_EA0_RD_UNCACHED_32B TCC_REQ"; do
  tag=$(echo $set | tr ' ' '+')
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/pmc_$tag -o pmc -- $R/tools/micro/adam_stream.bin 26 > $R/$O/pmc_$tag.log 2>&1)
  f=$(find $O/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a $O/counters.txt
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0]
    if 'mix' in k or 'adam_k<0>' in k or 'adam_k<1>' in k:
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in sorted(acc.items()):
    print(k[-60:], {c: round(sum(v[-10:])/len(v[-10:]),1) for c,v in d.items()}, 'n=',len(next(iter(d.values()))))
PY
done
