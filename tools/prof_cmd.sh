#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <script.py> [args...]  -- rocprofv3 kernel-trace stats of a tools/ script -> top kernels
tag=$1; script=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o p -- python $GRAFT_REPO_ROOT/$script "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print(f"{r['Name'].split('(')[0][-80:]:80s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
