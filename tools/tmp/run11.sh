python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "token_major_maps" 2>&1 | tail -2
for v in 1 0; do XFM_TOKEN_SS2D=$v python bench.py --steps 20 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_tok$v.json; done
