#!/usr/bin/env python
"""Which Python call sites make the framework copy / cast / cat kernels of one XFMamba-T forward pass?  (monkeypatched
Tensor.contiguous / .to / .float / .bfloat16 / torch.cat / F.pad / Tensor.permute-free; eager, batch 32, autocast bf16)"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LOG = collections.Counter()
BYTES = collections.Counter()


def site():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "xfmamba_amd" in f.filename and "copysites" not in f.filename:
            return f"{os.path.basename(f.filename)}:{f.lineno} {f.line[:70]}"
    return "?"


def main():
    from xfmamba_amd import _lib
    from xfmamba_amd.net_fusionmamba import TwoViewXFMambaTop
    _lib.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(42)
    model = TwoViewXFMambaTop(in_channels=1, outputs=2, type="tiny").to(dev).train()
    B = 32
    xa = torch.randn(B, 1, 224, 224, device=dev)
    xb = torch.randn(B, 1, 224, 224, device=dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        model(xa, xb).float().sum().backward()
    on = [False]
    oc, oto, ocat = torch.Tensor.contiguous, torch.Tensor.to, torch.cat
    ofl, obf = torch.Tensor.float, torch.Tensor.bfloat16

    def contiguous(self, *a, **k):
        if on[0] and self.is_cuda and not self.is_contiguous():
            key = ("contiguous", tuple(self.shape), tuple(self.stride()), str(self.dtype), site())
            LOG[key] += 1
            BYTES[key] += self.numel() * self.element_size() * 2
        return oc(self, *a, **k)

    def to(self, *a, **k):
        r = oto(self, *a, **k)
        if on[0] and self.is_cuda and r.data_ptr() != self.data_ptr():
            key = ("to", tuple(self.shape), str(self.dtype) + "->" + str(r.dtype), "", site())
            LOG[key] += 1
            BYTES[key] += self.numel() * (self.element_size() + r.element_size())
        return r

    def cat(ts, *a, **k):
        r = ocat(ts, *a, **k)
        if on[0] and r.is_cuda:
            key = ("cat", tuple(r.shape), str(r.dtype), "", site())
            LOG[key] += 1
            BYTES[key] += r.numel() * r.element_size() * 2
        return r

    def fl(self, *a, **k):
        r = ofl(self, *a, **k)
        if on[0] and self.is_cuda and self.dtype != torch.float32:
            key = ("float", tuple(self.shape), str(self.dtype), "", site())
            LOG[key] += 1
            BYTES[key] += self.numel() * (self.element_size() + 4)
        return r

    torch.Tensor.contiguous, torch.Tensor.to, torch.cat, torch.Tensor.float = contiguous, to, cat, fl
    on[0] = True
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(xa, xb)
    nf = sum(LOG.values())
    if "--fwd-only" in sys.argv:
        on[0] = False
    out.float().sum().backward()                            # (the Python backward of the custom nodes is patched too)
    on[0] = False
    torch.Tensor.contiguous, torch.Tensor.to, torch.cat, torch.Tensor.float = oc, oto, ocat, ofl
    torch.cuda.synchronize()
    print(f"{nf} copying calls in the forward pass, {sum(LOG.values()) - nf} more in the Python backward of the custom nodes, "
          f"{sum(BYTES.values()) / 1e6:.1f} MB moved (autograd's own mirror copies not counted)")
    for key, n in sorted(LOG.items(), key=lambda kv: -BYTES[kv[0]])[:60]:
        print(f"{BYTES[key] / 1e6:8.2f} MB n={n:2d} {key[0]:10s} {str(key[1]):24s} {str(key[2])[:34]:34s} {key[4]}")


if __name__ == "__main__":
    main()
