cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "dt_proj or ss2d" 2>&1 | tail -2
python bench.py --no-kernel-timer --no-cpu-baseline --steps 20 --model small 2>&1 | tail -1 | cut -c1-200
for k in 1 2 3 4 5 6; do
XFM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 2 --no-kernel-timer --no-cpu-baseline 2>&1 | grep -a "diverged\|^{" | cut -c1-100 | sed 's/^{.*/OK/' | head -1
done
