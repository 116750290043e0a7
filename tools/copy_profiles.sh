#!/bin/bash
# copy the summaries of gpurun_out/measure_<round>/ (tools/measure_round.sh) into profiles/<round>_* (tracked)
R=${1:-r04}
O=gpurun_out/measure_$R
tail -1 $O/bench_line.json > profiles/${R}_bench_line.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) profiles/${R}_bench_step_kernel_stats.csv
cp $O/traffic_pmc.csv profiles/${R}_bench_traffic_pmc.csv
cp $O/traffic.json profiles/${R}_traffic.json
cp $O/mfma_util.csv profiles/${R}_mfma_util.csv
cp $O/mfma_util.json profiles/${R}_mfma_util.json
cp $O/steady_kernel_stats.csv profiles/${R}_steady_state_kernel_stats.csv
cp $O/steady_families.txt profiles/${R}_steady_state_families.txt
cp $O/sq_lean.csv profiles/${R}_ss2d_wide_sq_counters.csv
cp $O/sq_chan.csv profiles/${R}_ss2d_chan_sq_counters.csv
ls -la profiles/${R}_*
