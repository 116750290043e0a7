#!/bin/bash
# kernel durations of tools/tilebench.py in launch order (own kernel vs library, shape by shape)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_tile
rm -rf $O && mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 tools/tilebench.py > $O/log.txt 2>&1
grep -v "^W\|^E" $O/log.txt | tail -14
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_tile/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# consecutive runs of the same kernel name -> one line (count, median duration)
out = []; cur = None
for r in rows:
    n = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if cur and cur[0] == n: cur[1].append(d)
    else:
        cur = [n, [d]]; out.append(cur)
lines = []
for n, ds in out:
    if len(ds) < 3 and 'xfm' not in n: continue
    if 'tile_gemm' in n and len(ds) == 50:                 # conv section: fwd, then the four dgrad parity classes, repeated
        per = [sorted(ds[i::5])[len(ds) // 10] for i in range(5)]
        lines.append("conv fwd %.1f us  dgrad classes %s = %.1f us" % (per[0], ["%.1f" % v for v in per[1:]], sum(per[1:])))
        continue
    ds = sorted(ds)
    lines.append(f"{len(ds):4d} x {ds[len(ds)//2]:9.1f} us  {n[:110]}")
open('gpurun_out/prof_tile/summary.txt', 'w').write("\n".join(lines) + "\n")
print("\n".join(l for l in lines if 'xfm' in l or 'Cijk' in l or 'igemm' in l or 'Sp3Asm' in l or l.startswith('conv')))
PY
find $O -name "*kernel_trace.csv" -delete
