run() { echo "== $*"; env "$@" XFM_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timer 2>&1 | grep -E "^\{|loss diverged|Error" | head -2 | cut -c1-200; }
run XFM_SHALLOW_KERNEL=0
run XFM_SS2D_YTOK=0
run XFM_TILED_LINEAR=0
run XFM_SHALLOW_KERNEL=0 XFM_SS2D_YTOK=0 XFM_TILED_LINEAR=0
echo "== dp-cut -1"; XFM_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timer --dp-cut -1 2>&1 | grep -E "^\{|loss diverged|Error" | head -2 | cut -c1-200
echo "== no-graph"; XFM_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timer --no-graph 2>&1 | grep -E "^\{|loss diverged|Error" | head -2 | cut -c1-200
