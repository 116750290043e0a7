#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_wgrad
rm -rf $O && mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 tools/wgradbench.py > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_wgrad/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = []; cur = None
for r in rows:
    n = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'Fill' in n or 'fill' in n: continue
    if cur and cur[0] == n: cur[1].append(d)
    else:
        cur = [n, [d]]; out.append(cur)
# per shape: own kernel (5 consecutive launches), then library GEMM + reduce alternating (5 pairs)
i = 0
while i < len(out):
    n, ds = out[i]
    if 'wgrad' in n and 'kernel' in n:
        own = sorted(ds)[len(ds) // 2]
        lib = []
        j = i + 1
        while j < len(out) and not ('wgrad' in out[j][0] and 'kernel' in out[j][0]):
            lib += out[j][1]; j += 1
        per = sum(lib) / 5.0 if lib else 0.0
        print(f"own {own:7.1f} us   library gemm+sum {per:7.1f} us   {n[30:62]}")
        i = j
    else:
        i += 1
PY
find $O -name "*kernel_trace.csv" -delete
