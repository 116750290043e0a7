for c in 5d57340 0d880eb e8b0367; do
  echo "=== $c"
  cd $GRAFT_REPO_ROOT/tmp_$c && make -s -C xfmamba_amd/csrc -j32 2>&1 | grep -E " error" ; make -s -C oracle 2>&1 | tail -1
  XFM_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timer 2>&1 | grep -E "^\{|loss diverged|Error" | head -2 | cut -c1-160
done
