python -m pytest tests/test_hip_chan.py -x -q -m gpu -k "token_major" 2>&1 | tail -8
python -m pytest tests/test_hip_model.py -x -q -m gpu -k "tiny" 2>&1 | tail -8
for v in "1 1" "0 1" "1 0" "0 0"; do set -- $v; XFM_SS2D_YTOK=$1 XFM_TILED_LINEAR=$2 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timer 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ytok $1 tiled $2', d['value'], d['ms_per_step'])"; done
