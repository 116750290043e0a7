cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "dt_proj" 2>&1 | tail -2
