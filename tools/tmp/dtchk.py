import sys, os
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/xfmamba_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from xfmamba_amd import _lib
lib = _lib.lib()
for (B, D, R, H) in [(32, 192, 12, 56), (32, 192, 16, 56), (32, 192, 6, 28), (32, 192, 6, 12), (32, 192, 6, 16), (32, 128, 6, 56), (32, 128, 8, 56)]:
    L = H * H
    g = torch.Generator().manual_seed(D * R)
    ddts = torch.randn(B, 4, D, L, generator=g).bfloat16()
    xr = torch.randn(B, 4, R, L, generator=g).bfloat16()
    w = (torch.randn(4, D, R, generator=g) * R ** -0.5).bfloat16()
    dxr_ref = torch.einsum("kdr,bkdl->bkrl", w.float().cuda(), ddts.float().cuda())
    dw_ref = torch.einsum("bkdl,bkrl->kdr", ddts.float().cuda(), xr.float().cuda())
    dd, xd, wd = ddts.cuda(), xr.cuda(), w.cuda()
    dxr = torch.full((B, 4, R, L), float("nan"), dtype=torch.bfloat16, device="cuda")
    dw = torch.zeros(4, D, R, device="cuda")
    rc = lib.xfm_ss2d_dt_proj_bwd_mfma(dd.data_ptr(), xd.data_ptr(), wd.data_ptr(), dxr.data_ptr(), dw.data_ptr(), B, D, R, L, _lib.stream_ptr())
    torch.cuda.synchronize()
    e1 = float((dxr.float() - dxr_ref).abs().max()) / float(dxr_ref.abs().max())
    e2 = float((dw - dw_ref).abs().max()) / float(dw_ref.abs().max())
    print((B, D, R, H), "rc", rc, "dxr err", f"{e1:.2e}", "nan", int(torch.isnan(dxr.float()).sum()), "dw err", f"{e2:.2e}")
