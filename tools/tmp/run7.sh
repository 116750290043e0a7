python -m pytest tests/test_hip_chan.py -x -q -m gpu -k "swap" 2>&1 | tail -15
python -m pytest tests/test_hip_model.py -x -q -m gpu -k "shallow or golden" 2>&1 | tail -5
