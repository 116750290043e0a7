#!/bin/bash
# rocprofv3 kernel durations of tools/chanbench.py (channel-lane vs lean SS2D kernels), run through gpurun
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_chan
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o chan -- python3 tools/chanbench.py "$@" > $O/chanbench.log 2>&1
find $O -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_chan/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:30]:
    print(r['Name'][:100], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
