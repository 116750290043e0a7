cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in "" "--no-miopen-find" "" "--no-miopen-find"; do s=$(date +%s); python bench.py --steps 50 --no-kernel-timer --no-cpu-baseline $v 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('find' if '$v'=='' else 'nofind', j['value'], j['ms_per_step'])"; echo "wall $(( $(date +%s) - s )) s"; done
