cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 2300 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/measure_round.sh r05 > gpurun_out/measure_r05.log 2>&1
tail -3 gpurun_out/measure_r05.log
