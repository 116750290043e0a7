python -m pytest tests/test_hip_chan.py -x -q -m gpu -k "matches_oracle_chain" 2>&1 | tail -4
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "dt_proj or proj_core" 2>&1 | tail -3
python -m pytest tests/test_hip_model.py -x -q -m gpu -k "batch32_training" 2>&1 | tail -6
python tools/leanbench.py --only "T s" 2>&1 | tail -2
