cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "ss2d" 2>&1 | tail -2
for d in 0; do
  O=gpurun_out/prof_l3_$d; rm -rf $O; mkdir -p $O
  XFM_L3_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $O -o l -- python3 tools/leanbench.py --only "T s" > $O/log.txt 2>&1
  python3 - $d <<'PY'
import csv, glob, sys
d=sys.argv[1]
f = glob.glob(f'gpurun_out/prof_l3_{d}/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if 'l3_bwd' in r['Name'] or 'l3_fwd' in r['Name']: print('dbg', d, r['Name'][10:60], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
  rm -rf $O
done
