#!/bin/bash
# rocprofv3 kernel stats of the bench step (eager launches so every kernel is attributed), top kernels by total time
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_step
rm -rf $O && mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O -o s -- python3 bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-kernel-timer "$@" > $O/log.txt 2>&1
tail -c 300 $O/log.txt
python3 - <<'PY'
# Steady-state window only: MIOpen's convolution search (naive / ck / igemm candidates, seconds of kernels) runs inside
# the first steps of every fresh process.  adam_multi_kernel runs once per step: keep what lies between the ends of the
# 7th- and 1st-from-last optimizer launches = the last 6 steps.
import csv, glob, collections
f = glob.glob('gpurun_out/prof_step/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
adam = [e for s, e, n in rows if 'adam_multi_kernel' in n or 'multi_tensor_apply_kernel' in n and 'Adam' in n]
NS = 6
lo, hi = adam[-NS - 1], adam[-1]
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in rows:
    if s >= lo and e <= hi:
        tot[n] += e - s; cnt[n] += 1
with open('gpurun_out/prof_step/s_kernel_stats.csv', 'w') as out:
    w = csv.writer(out)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage'])
    T = sum(tot.values())
    for n, t in tot.most_common():
        w.writerow([n, cnt[n], t, t / cnt[n], 100.0 * t / T])
# launch order of the last step (name, start offset, duration): which kernels sit between which
with open('gpurun_out/prof_step/s_last_step_order.csv', 'w') as out:
    w = csv.writer(out)
    w.writerow(['Index', 'Name', 'StartUs', 'DurationUs'])
    i = 0
    for s, e, n in rows:
        if s >= adam[-2] and e <= adam[-1]:
            w.writerow([i, n[:140], round((s - adam[-2]) / 1e3, 1), round((e - s) / 1e3, 1)])
            i += 1
print("steady-state window: %d steps, wall %.2f ms/step, kernel time %.2f ms/step" % (NS, (hi - lo) / 1e6 / NS, T / 1e6 / NS))
for n, t in tot.most_common(45):
    print(f"{n[:95]:95s} {cnt[n] / NS:6.1f} {t / 1e6 / NS:8.3f} ms/step {t / cnt[n] / 1e3:8.1f} us")
PY
find $O -name "*kernel_trace.csv" -delete
