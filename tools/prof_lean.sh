cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_lean
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o l -- python3 tools/leanbench.py > $O/log.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
cat $O/log.txt | grep -v "^W\|^E" | tail -8
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_lean/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    if 'xfm::' in r['Name']: print(r['Name'][:110], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
