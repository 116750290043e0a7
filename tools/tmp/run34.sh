cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for k in 1 2 3 4 5 6 7 8 9 10 11 12; do
echo "== run $k"; XFM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 2 --no-kernel-timer --no-cpu-baseline 2>&1 | grep -a "non-finite\|diverged\|^{" | cut -c1-300 | sed 's/^{.*/OK/' | head -2
done
