cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "dt_proj" 2>&1 | tail -2
for op in dtbwd0 dtbwd1; do
python3 tools/stress2.py $op 300 & P1=$!
python3 tools/stress2.py $op 300 & P2=$!
wait $P1; wait $P2
done
