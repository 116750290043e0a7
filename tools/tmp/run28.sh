cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_dp_gloo.py -x -q -m gpu -k "two_ranks_on_one_gpu_run" 2>&1 | tail -60
