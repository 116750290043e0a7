#!/bin/bash
# SQ counters of the fused scan kernels at the bench shapes (one rocprofv3 --pmc pass over tools/kbench.py);
# writes gpurun_out/sq_r01/summary.csv (copied to profiles/rNN_ss2d_sq_counters.csv).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
# SQTOOL=chanbench profiles tools/chanbench.py (channel-lane kernels) instead of tools/kbench.py --only $KB
O=gpurun_out/sq_r01
rm -rf $O && mkdir -p $O
if [ "${SQTOOL:-kbench}" = "chanbench" ]; then
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/p -o sq -- python3 tools/chanbench.py ${CHANARGS:-} > $O/log.txt 2>&1
elif [ "${SQTOOL:-kbench}" = "leanbench" ]; then
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/p -o sq -- python3 tools/leanbench.py ${LEANARGS:-} > $O/log.txt 2>&1
else
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/p -o sq -- python3 tools/kbench.py --only ${KB:-ss2d} > $O/log.txt 2>&1
fi
tail -3 $O/log.txt
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob('gpurun_out/sq_r01/p/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.Counter()); n = collections.Counter()
for r in csv.DictReader(open(f)):
    m = re.search(r'xfm::((?:\w+::)*\w+_kernel<[^>]*>)', r['Kernel_Name'])
    if not m: continue
    k = m.group(1) + ' grid=' + r.get('Grid_Size', r.get('Grid_Size_X', ''))
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES': n[k] += 1
with open('gpurun_out/sq_r01/summary.csv', 'w') as out:
    out.write('kernel,launches,waves,valu_per_wave,salu_per_wave,lds_per_wave,wave_cycles_per_wave,valu_active_frac,wait_any_frac,wait_inst_frac\n')
    for k, c in sorted(acc.items()):
        w = c['SQ_WAVES'] / max(n[k], 1)
        wc = c['SQ_WAVE_CYCLES'] / max(c['SQ_WAVES'], 1)
        out.write(f"\"{k}\",{n[k]},{w:.0f},{c['SQ_INSTS_VALU']/c['SQ_WAVES']:.0f},{c['SQ_INSTS_SALU']/c['SQ_WAVES']:.0f},{c['SQ_INSTS_LDS']/c['SQ_WAVES']:.0f},{wc:.0f},{c['SQ_ACTIVE_INST_VALU']/max(c['SQ_WAVE_CYCLES'],1):.3f},{c['SQ_WAIT_ANY']/max(c['SQ_WAVE_CYCLES'],1):.3f},{c['SQ_WAIT_INST_ANY']/max(c['SQ_WAVE_CYCLES'],1):.3f}\n")
print(open('gpurun_out/sq_r01/summary.csv').read())
PY
find $O -name "*counter_collection.csv" -size +5M -delete; find $O -name "*kernel_trace.csv" -delete
