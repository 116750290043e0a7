"""LayerNorm2d forward / backward timing at the four stage shapes of XFMamba-T (batch 64 = 2 views x 32)."""
import sys
import torch
sys.path.insert(0, ".")
from xfmamba_amd.layernorm2d import layernorm2d_fn

dev = torch.device("cuda:0")
for (B, C, H) in [(64, 96, 56), (64, 192, 28), (64, 384, 14), (64, 768, 7), (32, 1536, 7)]:
    x = torch.randn(B, C, H, H, device=dev, dtype=torch.float32, requires_grad=True)      # (the trunk: fp32 planes in, bf16 out)
    w = torch.ones(C, device=dev, requires_grad=True)
    b = torch.zeros(C, device=dev, requires_grad=True)
    gy = torch.randn(B, C, H, H, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        y = layernorm2d_fn(x, w, b, 1e-5, torch.bfloat16)
        y.backward(gy)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n = 20
    torch.cuda.synchronize()
    e[0].record()
    ys = [layernorm2d_fn(x, w, b, 1e-5, torch.bfloat16) for _ in range(n)]
    e[1].record()
    for y in ys:
        y.backward(gy)
    e[2].record()
    torch.cuda.synchronize()
    nb = x.numel() * 2            # (bf16-equivalent bytes: the rates printed are indicative only)
    f, bw = e[0].elapsed_time(e[1]) / n * 1e3, e[1].elapsed_time(e[2]) / n * 1e3
    print(f"{B}x{C}x{H}x{H}: fwd {f:7.1f} us ({2 * nb / f / 1e6:6.2f} TB/s)  bwd {bw:7.1f} us ({3 * nb / bw / 1e6:6.2f} TB/s)")
