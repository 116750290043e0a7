import torch, time
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = "cuda"
def bench(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for (Ci, Co, H) in [(96, 192, 56), (192, 384, 28), (384, 768, 14), (3, 48, 224), (48, 96, 112)]:
    for cl in (False, True):
        x = torch.randn(64, Ci, H, H, device=dev, dtype=torch.bfloat16)
        w = torch.randn(Co, Ci, 3, 3, device=dev, dtype=torch.bfloat16) * 0.05
        b = torch.randn(Co, device=dev, dtype=torch.bfloat16)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last); w = w.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(); w.requires_grad_(); b.requires_grad_()
        y = F.conv2d(x, w, b, stride=2, padding=1)
        gy = torch.randn_like(y)
        def f():
            y = F.conv2d(x, w, b, stride=2, padding=1)
            y.backward(gy)
        def ff():
            with torch.no_grad(): F.conv2d(x, w, b, stride=2, padding=1)
        print(Ci, Co, H, "channels_last" if cl else "contiguous", "out strides", y.stride(), "is_cl", y.is_contiguous(memory_format=torch.channels_last),
              "fwd us", round(bench(ff), 1), "fwd+bwd us", round(bench(f), 1), flush=True)
