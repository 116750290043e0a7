#!/bin/bash
# rocprofv3 kernel stats of the bench step (eager launches so every kernel is attributed), top kernels by total time
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_step
rm -rf $O && mkdir -p $O
# (a first, unprofiled run fills MIOpen's per-user find cache: on a fresh box the convolution search otherwise lands in
#  the profile -- seconds of naive_conv / ck search kernels)
python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timer "$@" > $O/warm.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timer "$@" > $O/log.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
tail -c 300 $O/log.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_step/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms:", tot/1e6)
for r in rows[:45]:
    print(f"{r['Name'][:95]:95s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
PY
