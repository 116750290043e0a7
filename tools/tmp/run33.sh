cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python3 tools/nanpoison.py 2>&1 | grep -a "^step\|Error\|error\|assert" | head
XFM_PHASED=1 timeout 900 python3 tools/nanpoison.py 2>&1 | grep -a "^step\|Error\|error\|assert" | head
