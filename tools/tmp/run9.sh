python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "token_major_maps" 2>&1 | tail -8
python -m pytest tests/test_hip_model.py -x -q -m gpu -k "tiny" 2>&1 | tail -6
for v in 1 0; do XFM_TOKEN_SS2D=$v python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timer 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('token_ss2d $v', d['value'], d['ms_per_step'])"; done
