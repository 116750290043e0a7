cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "conv3x3_stride2 or wgrad" 2>&1 | tail -5
python3 tools/convprobe.py 2>&1 | tail -4
