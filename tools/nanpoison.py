#!/usr/bin/env python
"""Find reads of uninitialised device memory: fill the caching allocator's pool with NaN bit patterns (bf16 / fp32 NaN alike), then
run eager training steps of the bench model and report which gradients / parameters come out non-finite.  A kernel that relies on
`torch.empty` contents -- or multiplies padding it never wrote by zero -- is invisible in a single process (fresh memory is zeros or
stale finite data) and shows up only when another process's leftovers are in the blocks: this makes it deterministic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def poison(gb=12):
    blocks = []
    for sz in [16 << 30] * gb + [1 << 26] * 64 + [1 << 22] * 256 + [1 << 18] * 512 + [1 << 14] * 1024 + [4096] * 2048 + [512] * 4096:
        t = torch.empty(sz // 4, dtype=torch.int32, device="cuda")
        t.fill_(-1)                                   # 0xffffffff: NaN as fp32, and as both bf16 halves
        blocks.append(t)
    torch.cuda.synchronize()
    del blocks                                        # back to the caching allocator, contents intact
    probe = [torch.empty(n, device="cuda") for n in (100, 100000, 50_000_000)]
    assert all(bool(torch.isnan(t).all()) for t in probe), "the allocator did not hand the poisoned blocks back"


def main():
    from xfmamba_amd.net_fusionmamba import TwoViewXFMambaTop
    from xfmamba_amd.optim import FusedAdam
    from xfmamba_amd import deferred
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    torch.manual_seed(0)
    m = TwoViewXFMambaTop(in_channels=1, outputs=2, type="tiny").cuda().train()
    opt = FusedAdam(m.parameters(), lr=1e-4)
    xa, xb = torch.randn(B, 1, 224, 224, device="cuda"), torch.randn(B, 1, 224, 224, device="cuda")
    lab = torch.randint(0, 2, (B,), device="cuda")
    names = {id(p): n for n, p in m.named_parameters()}
    if os.environ.get("XFM_DEFER", "1") == "1":
        deferred.defer_partial_sums(True)
    phased = None
    if os.environ.get("XFM_PHASED", "0") == "1":      # the data-parallel step's two backward pieces (world size 1: no collective)
        from xfmamba_amd.dp import PhasedGrads
        from xfmamba_amd.proj import WgradArena, set_wgrad_arena
        phased = PhasedGrads(m, wire_dtype=torch.bfloat16)
        m.mamba_feature_extrac.cut_after = 1
        arena = WgradArena(m.parameters())
        set_wgrad_arena(arena)
    for step in range(4):
        poison()
        for p in m.parameters():
            p.grad = None
        if phased is not None:
            arena.zero()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(m(xa, xb).float(), lab)
        if phased is None:
            loss.backward()
            deferred.flush()
            torch.cuda.synchronize()
            bad = [names[id(p)] for p in m.parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
            print(f"step {step}: loss {float(loss):.4f} finite {bool(torch.isfinite(loss))}; non-finite gradients: {len(bad)}", bad[:12])
            opt.step()
        else:
            phased.backward_late(loss, m.mamba_feature_extrac.cut_tensor)
            phased.backward_early()
            torch.cuda.synchronize()
            bad = [names[id(p)] for i in (0, 1) for p, v in zip(phased.pieces[i], phased.views[i]) if not bool(torch.isfinite(v.float()).all())]
            print(f"step {step} (two pieces): loss {float(loss):.4f}; non-finite wire slots: {len(bad)}", bad[:12])
            opt.step(grads=phased.grads(), grad_scale=phased.grad_scale)


if __name__ == "__main__":
    main()
