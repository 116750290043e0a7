cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for pad in 0 30000; do for d in 0 48; do
  O=gpurun_out/prof_l3_$d; rm -rf $O; mkdir -p $O
  XFM_L3_LDS_PAD=$pad XFM_L3_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d $O -o l -- python3 tools/leanbench.py --only "T s" > $O/log.txt 2>&1
  python3 - $d $pad <<'PY'
import csv, glob, sys
d=sys.argv[1]
f = glob.glob(f'gpurun_out/prof_l3_{d}/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if 'l3_bwd' in r['Name'] or 'l3_fwd' in r['Name']: print('pad', sys.argv[2], 'dbg', d, r['Name'][10:60], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
  rm -rf $O
done; done
