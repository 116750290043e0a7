#!/bin/bash
# rocprofv3 kernel durations of tools/mlpbench.py (fused GELU products vs the three-node chain), through gpurun
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for f in ${MLPFUSED:-1 0}; do
  O=gpurun_out/prof_mlp_$f
  rm -rf $O && mkdir -p $O
  XFM_MLP_FUSED=$f rocprofv3 --kernel-trace --output-format csv -d $O -o m -- python3 tools/mlpbench.py > $O/log.txt 2>&1
  python3 - $f <<'PY'
import csv, glob, sys, collections
f = sys.argv[1]
fn = glob.glob(f'gpurun_out/prof_mlp_{f}/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# three shapes x 8 iterations: group kernels by (name, grid) and report the median duration and count per iteration
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'randn' in n or 'distribution' in n or 'Fill' in n: continue
    agg[(n[:90], r.get('Grid_Size', r.get('Grid_Size_X', '')))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0.0
for (n, g), d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if len(d) < 8: continue
    d.sort(); med = d[len(d) // 2]; per = len(d) / 8.0
    tot += med * per
    print(f"fused={f} {med:8.1f} us x{per:4.1f}  grid {g:>9s}  {n}")
print(f"fused={f} total per iteration (3 shapes): {tot:8.1f} us")
PY
  rm -rf $O
done
