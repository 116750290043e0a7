cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "replicated_channel" 2>&1 | tail -2
for w in 256 512 1024 2048; do echo wgs $w; XFM_GRAY_WGS=$w python3 tools/convprobe.py 2>&1 | grep "gray"; done
