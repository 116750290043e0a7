import sys, torch
sys.path.insert(0, '.')
from xfmamba_amd import _lib, fusion_vmamba as fv
B, D, HW, R = 32, 256, 96, 8
L, K, N = HW*HW, 4, 1
g = torch.Generator().manual_seed(0)
dev='cuda'
x = torch.randn(B, D, HW, HW, generator=g).to(dev).bfloat16().requires_grad_()
xw = (torch.randn(K, R + 2 * N, D, generator=g) * D ** -0.5).to(dev).requires_grad_()
dtw = (torch.randn(K, D, R, generator=g) * R ** -0.5).to(dev).requires_grad_()
Alog = torch.zeros(K * D, N).to(dev).requires_grad_()
Dp = torch.randn(K * D, generator=g).to(dev).requires_grad_()
bias = (0.1 * torch.rand(K, D, generator=g) - 4.0).to(dev).requires_grad_()
gy = torch.randn(B, D, L, device=dev)
for mode in ("fused", "unfused"):
    fv.SS2D_MODE = mode
    timer = _lib.KernelTimer(); _lib.set_timer(timer)
    for _ in range(3):
        y, _ = fv._ss2d_core(x, xw, dtw, Alog, Dp, bias)
        y.backward(gy)
    torch.cuda.synchronize(); _lib.set_timer(None)
    print(mode, "  ".join(f"{k}={v['avg_us']:.0f}us" for k, v in timer.summary().items()))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        y, _ = fv._ss2d_core(x, xw, dtw, Alog, Dp, bias); y.backward(gy)
    e.record(); e.synchronize()
    print(mode, "fwd+bwd per call ms:", s.elapsed_time(e)/3)
