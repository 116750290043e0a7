cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for k in 1 2; do
echo "== HEAD run $k"; XFM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 2 --no-kernel-timer --no-cpu-baseline 2>&1 | grep -a "loss diverged\|^{" | cut -c1-100 | head -1
done
cd tools/tmp/wt_c6
for k in 1 2; do
echo "== c6b552b run $k"; XFM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 2 --no-kernel-timer --no-cpu-baseline 2>&1 | grep -a "loss diverged\|^{" | cut -c1-100 | head -1
done
