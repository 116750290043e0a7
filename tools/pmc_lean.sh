# HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, separate passes) and durations of the scan kernels at the trunk shapes of
# XFMamba-T: bash tools/pmc_lean.sh            wide-map kernels (tools/leanbench.py)
#            TOOL=chanbench bash tools/pmc_lean.sh   channel-lane / deep kernels (tools/chanbench.py)
# (through gpurun; set XFM_* tuning variables in the environment to A/B them)
TOOL=${TOOL:-leanbench}
ONLY="T s"; PAT="ss2d_l3"
if [ "$TOOL" = chanbench ]; then ONLY="T"; PAT="ss2dc_|deep_bwd"; fi
export PAT
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_lean
rm -rf $O && mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o l -- python3 tools/$TOOL.py --only "$ONLY" > $O/log.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o p -- python3 tools/$TOOL.py --only "$ONLY" > $O/pmc_$C.log 2>&1
done
find $O -name "*kernel_trace.csv" -delete
python3 tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv") $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv") 1 $O/traffic_pmc.csv $O/traffic.json > /dev/null
python3 - <<'PY'
import csv, glob, json
t = json.load(open('gpurun_out/pmc_lean/traffic.json'))
f = glob.glob('gpurun_out/pmc_lean/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    import os, re
    if re.search(os.environ["PAT"], n):
        key = n.split('xfm::')[1].split('(')[0]
        print(f"{key:40s} {float(r['AverageNs'])/1e3:8.1f} us   HBM {t.get(key, {}).get('hbm_bytes_per_launch', 0)/1e6:8.1f} MB per launch")
PY
