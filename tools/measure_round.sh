#!/bin/bash
# End-of-round measurement on one MI355X (run through gpurun): default bench line, rocprofv3 kernel stats of the same
# command, the two PMC passes (FETCH_SIZE / WRITE_SIZE, own runs, kernel-trace only) for the roofline's `traffic`, and one
# PMC pass for the matrix-core utilisation of the GEMM kernels.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=${1:-r03}
O=gpurun_out/measure_$R
rm -rf $O && mkdir -p $O
python3 bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 600 $O/bench_line.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline > $O/stats.log 2>&1
find $O/stats -name "*kernel_trace.csv" -delete
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-miopen-find --no-graph --no-kernel-timer > $O/pmc_$C.log 2>&1
  find $O/pmc_$C -name "*kernel_trace.csv" -delete
done
python3 tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv") $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv") 2 $O/traffic_pmc.csv $O/traffic.json | head -5
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-miopen-find --no-graph --no-kernel-timer > $O/pmc_mfma.log 2>&1
find $O/pmc_mfma -name "*kernel_trace.csv" -delete
python3 tools/pmc_mfma.py $(find $O/pmc_mfma -name "*counter_collection.csv") $O/mfma_util.csv $O/mfma_util.json | head -30
find $O -name "*counter_collection.csv" -size +20M -delete
# steady-state step profile (the last 6 of 12 eager steps: MIOpen's per-process convolution search excluded) + families
bash tools/prof_step.sh --no-graph > $O/steady.txt 2>&1
cp gpurun_out/prof_step/s_kernel_stats.csv $O/steady_kernel_stats.csv
python3 tools/stepcat.py $O/steady_kernel_stats.csv > $O/steady_families.txt
cat $O/steady_families.txt
# SQ counters of the scan kernels at the bench shapes
SQTOOL=leanbench bash tools/sq_counters.sh > $O/sq_lean.txt 2>&1; cp gpurun_out/sq_r01/summary.csv $O/sq_lean.csv
SQTOOL=chanbench bash tools/sq_counters.sh > $O/sq_chan.txt 2>&1; cp gpurun_out/sq_r01/summary.csv $O/sq_chan.csv
find $O -name "*counter_collection.csv" -size +5M -delete
du -sh $O
