cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for cfg in "XFM_CONV_OWN=0" "XFM_DTPROJ_MERGED=0" "XFM_SS2D_L3=0" "XFM_CONV_GRAY=0" "XFM_CONV_WGRAD_X=0" "XFM_CONV_OWN_MIN_C=100000"; do
echo "== $cfg"; env $cfg python bench.py --no-kernel-timer --no-cpu-baseline --steps 6 --warmup 3 --model small 2>&1 | grep -a "diverged\|^{" | cut -c1-80 | sed 's/^{.*/OK/' | head -1
done
