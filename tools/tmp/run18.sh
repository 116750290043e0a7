cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "gemm2 or tiled or mlp" 2>&1 | tail -3
echo BK32; python3 tools/gemm3probe.py 2>&1 | grep "^T"
echo BK64; XFM_GEMM3_BK=64 python3 tools/gemm3probe.py 2>&1 | grep "^T"
