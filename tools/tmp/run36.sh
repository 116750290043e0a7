cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/stress
(timeout 1500 python -m pytest tests/test_hip_ops.py tests/test_hip_chan.py -q -m gpu -p no:cacheprovider > gpurun_out/stress/a.txt 2>&1) & P1=$!
(timeout 1500 python -m pytest tests/test_hip_chan.py tests/test_hip_ops.py -q -m gpu -p no:cacheprovider > gpurun_out/stress/b.txt 2>&1) & P2=$!
wait $P1; wait $P2
tail -4 gpurun_out/stress/a.txt; tail -4 gpurun_out/stress/b.txt
grep -h "^FAILED" gpurun_out/stress/a.txt gpurun_out/stress/b.txt | head -20
