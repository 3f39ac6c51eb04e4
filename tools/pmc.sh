#!/bin/bash
# usage: tools/pmc.sh <tag> <script.py> <counters...>   -- rocprofv3 PMC pass (kernel-trace only) -> per-kernel mean counters
tag=$1; script=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/$script > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    if 'rlhip' not in k: continue
    print(k[-80:], {c: round(sum(v)/len(v),1) for c,v in d.items()}, 'n=',len(next(iter(d.values()))))
PY
