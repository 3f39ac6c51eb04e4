#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]  -- rocprofv3 kernel-trace stats of bench.py into gpurun_out/prof_<tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o bench -- python $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
grep -v "^W2026\|^E2026\|amdgpu.ids" $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log | tail -1 | cut -c1-300
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"{r['Name'].split('(')[0][-70:]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
