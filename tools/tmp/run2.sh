python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "dt_proj_inside" 2>&1 | tail -3
for f in 1 0; do XFM_SS2D_DT_FUSED=$f python tools/leanbench.py --only "T s" 2>&1 | tail -2; done
