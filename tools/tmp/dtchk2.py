import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from xfmamba_amd import _lib
lib = _lib.lib()
B, D, R, H = [int(v) for v in sys.argv[1:5]]
L = H * H
g = torch.Generator().manual_seed(D * R)
dd = torch.randn(B, 4, D, L, generator=g).bfloat16().cuda()
xd = torch.randn(B, 4, R, L, generator=g).bfloat16().cuda()
wd = (torch.randn(4, D, R, generator=g) * R ** -0.5).bfloat16().cuda()
dxr = torch.full((B, 4, R, L), float("nan"), dtype=torch.bfloat16, device="cuda")
dw = torch.zeros(4, D, R, device="cuda")
torch.cuda.synchronize()
print("before", (B, D, R, H), flush=True)
rc = lib.xfm_ss2d_dt_proj_bwd_mfma(dd.data_ptr(), xd.data_ptr(), wd.data_ptr(), dxr.data_ptr(), dw.data_ptr(), B, D, R, L, _lib.stream_ptr())
torch.cuda.synchronize()
print("after kernel rc", rc, "nan in dxr", int(torch.isnan(dxr.float()).sum()), flush=True)
dxr_ref = torch.einsum("kdr,bkdl->bkrl", wd.float(), dd.float())
print("dxr err", float((dxr.float() - dxr_ref).abs().max()) / float(dxr_ref.abs().max()), flush=True)
