python -m pytest tests/test_hip_chan.py -x -q -m gpu 2>&1 | tail -2
python -m pytest tests/test_hip_model.py -x -q -m gpu -k "deep" 2>&1 | tail -2
for d in 0 0; do python tools/chanbench.py --only "deep" 2>&1 | tail -1 | sed 's/.*ss2dc16_fwd=\([0-9.]*\)us  ss2dc16_bwd=\([0-9.]*\)us.*/fwd \1 bwd \2/'; done
