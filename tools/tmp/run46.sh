cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "gelu_inside or add_layernorm_rows" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_hip_model.py -x -q -m gpu 2>&1 | tail -2
for v in 1 0 1 0; do XFM_LN_GELU=$v python bench.py --steps 50 --no-kernel-timer --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('ln_gelu $v', j['value'], j['ms_per_step'])"; done
