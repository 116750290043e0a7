cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for c in "32 192 12 28" "32 192 6 28" "32 192 6 56" "32 192 12 56" "32 192 8 56" "32 96 6 56" "32 128 8 12" "32 96 12 56" "32 96 24 56"; do timeout 120 python3 tools/tmp/dtchk2.py $c 2>&1 | grep "dxr err"; done
for op in dtbwd0 dtbwd1; do
python3 tools/stress2.py $op 300 & P1=$!
python3 tools/stress2.py $op 300 & P2=$!
wait $P1; wait $P2
done
python bench.py --no-kernel-timer --no-cpu-baseline --steps 20 --model small 2>&1 | tail -1 | cut -c1-200
