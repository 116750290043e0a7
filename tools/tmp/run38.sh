cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
b() { python bench.py --no-kernel-timer --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
b --steps 50
b --steps 50
b --steps 30 --model small
b --steps 20 --model base --size 384 --batch 16
b --steps 20 --dtype fp32
b --steps 30 --fp8
b --steps 30 --drop-path 0
XFM_SS2D_DT_FUSED=1 python bench.py --no-kernel-timer --no-cpu-baseline --steps 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dt_fused=1', d['value'], d['ms_per_step'])"
XFM_TOKEN_SS2D=1 python bench.py --no-kernel-timer --no-cpu-baseline --steps 50 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('token_ss2d=1', d['value'], d['ms_per_step'])"
python tools/infer_bench.py 2>&1 | tail -3
