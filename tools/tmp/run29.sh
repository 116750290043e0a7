cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for cfg in "XFM_CONV_OWN=0" "XFM_CONV_GRAY=0 XFM_CONV_WGRAD_X=0" "XFM_CONV_GRAY=0" "XFM_CONV_WGRAD_X=0" "XFM_CONV_OWN_MIN_C=100000"; do
echo "== $cfg"; env $cfg XFM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 2 --no-kernel-timer --no-cpu-baseline 2>&1 | grep -a "loss diverged\|^{" | cut -c1-120 | head -3
done
