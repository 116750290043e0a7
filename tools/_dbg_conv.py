import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = "cuda"
def trial(name, fn, n=5):
    ref = [t.clone() for t in fn()]
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    res = []
    for i in range(n):
        g.replay(); torch.cuda.synchronize()
        res.append([(bool(torch.isfinite(o).all()), round(float((o.float() - r.float()).abs().max() / (r.float().abs().max() + 1e-9)), 4)) for o, r in zip(out, ref)])
    print(name, res, flush=True)
for cl in (False, True):
  for auto in (False, True):
    x = torch.randn(64, 96, 56, 56, device=dev)
    if cl: x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    x.requires_grad_()
    conv = torch.nn.Conv2d(96, 192, 3, 2, 1).to(dev)
    gy = torch.randn(64, 192, 28, 28, device=dev)
    def f():
        conv.zero_grad(set_to_none=True); x.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=auto):
            y = conv(x)
        y.float().backward(gy)
        return conv.bias.grad, conv.weight.grad, x.grad
    trial(f"conv cl={cl} autocast={auto}", f)
