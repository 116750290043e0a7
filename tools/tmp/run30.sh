cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
XFM_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 2 --no-kernel-timer --no-cpu-baseline 2>&1 | grep -av "^\s*$" | grep -a "rank0\|bench\]" | head -40
