python -m pytest tests/test_hip_chan.py -x -q -m gpu -k "16" 2>&1 | tail -2
for d in 0 4 63 0; do echo "dbg=$d"; XFM_DEEP_DBG=$d python tools/chanbench.py --only "deep" 2>&1 | tail -1 | sed 's/.*ss2dc16_fwd=\([0-9.]*\)us  ss2dc16_bwd=\([0-9.]*\)us.*/fwd \1 bwd \2/'; done
