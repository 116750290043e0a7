cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
cp tools/tmp/timing/libxfm_hip.so xfmamba_amd/libxfm_hip.so
for bk in 64 32; do for d in 0 256 512 1024 2048 768 1792 3840; do
echo "BK $bk dbg $d"; XFM_GEMM3_BK=$bk XFM_GEMM2_DBG=$d python3 tools/gemm3probe.py 5 2>&1 | grep "^T" | cut -c1-60
done; done
