cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/stress
(timeout 2000 python -m pytest tests/test_hip_model.py -q -m gpu -p no:cacheprovider > gpurun_out/stress/m1.txt 2>&1) & P1=$!
(timeout 2000 python -m pytest tests/test_hip_model.py -q -m gpu -p no:cacheprovider > gpurun_out/stress/m2.txt 2>&1) & P2=$!
wait $P1; wait $P2
tail -3 gpurun_out/stress/m1.txt; tail -3 gpurun_out/stress/m2.txt
grep -h "^FAILED" gpurun_out/stress/m1.txt gpurun_out/stress/m2.txt | head -20
