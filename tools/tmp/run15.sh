run() { echo "== $*"; env "$@" XFM_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timer 2>&1 | grep -E "^\{|loss diverged|Error" | head -2 | cut -c1-160; }
run XFM_DBG_OLD_TABLES=1
run XFM_DBG_INIT_ORDER=1
