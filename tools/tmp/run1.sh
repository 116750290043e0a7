set -x
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "dt_proj_inside or proj_core or xproj_core or fused_ss2d_matches or workspace_entry" 2>&1 | tail -15
for f in 1 0; do XFM_SS2D_DT_FUSED=$f python tools/leanbench.py --only "T s" 2>&1 | tail -3; done
XFM_SS2D_DT_FUSED=1 python tools/leanbench.py --only "S s0" 2>&1 | tail -3
