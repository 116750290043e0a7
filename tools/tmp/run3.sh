python -m pytest tests/test_hip_chan.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_hip_model.py -x -q -m gpu -k "deep" 2>&1 | tail -3
python tools/chanbench.py --only "deep" 2>&1 | tail -2
