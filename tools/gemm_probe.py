"""Probe: library GEMM time for the trunk's shapes with the weight stored (N, K) or transposed (K, N)."""
import torch


def t(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def main():
    dev = "cuda"
    bf = torch.bfloat16
    print("tokens GEMMs (Mlp): fwd y = x W^T, dgrad dx = dy W;  w: (N,K) storage, wt: (K,N) storage")
    for (T, K, N) in [(200704, 96, 384), (200704, 384, 96), (50176, 192, 768), (50176, 768, 192), (12544, 384, 1536),
                      (12544, 1536, 384), (3136, 768, 3072), (3136, 3072, 768)]:
        x = torch.randn(T, K, device=dev, dtype=bf)
        dy = torch.randn(T, N, device=dev, dtype=bf)
        w = torch.randn(N, K, device=dev, dtype=bf)
        wt = w.t().contiguous()
        a = t(lambda: torch.mm(x, w.t())); b = t(lambda: torch.mm(x, wt))
        c = t(lambda: torch.mm(dy, w)); d = t(lambda: torch.mm(dy, wt.t()))
        print(f"T={T:6d} K={K:4d} N={N:4d}  fwd: w {a:6.1f}  wt {b:6.1f}   dgrad: w {c:6.1f}  wt {d:6.1f}")
    print("batched projections (B=64): planes out y[b] = W x[b] (x tokens (B,L,K) or planes (B,K,L)); tokens out")
    for (L, K, M) in [(3136, 96, 192), (784, 192, 384), (196, 384, 768), (3136, 96, 96), (784, 192, 192), (196, 384, 384)]:
        B = 64
        xt = torch.randn(B, L, K, device=dev, dtype=bf)
        xp = torch.randn(B, K, L, device=dev, dtype=bf)
        w = torch.randn(M, K, device=dev, dtype=bf)
        wt = w.t().contiguous()
        r = {}
        r["tok->pl w"] = t(lambda: torch.bmm(w.unsqueeze(0).expand(B, M, K), xt.transpose(1, 2)))
        r["tok->pl wt"] = t(lambda: torch.bmm(wt.t().unsqueeze(0).expand(B, M, K), xt.transpose(1, 2)))
        r["pl->tok w"] = t(lambda: torch.bmm(xp.transpose(1, 2), w.t().unsqueeze(0).expand(B, K, M)))
        r["pl->tok wt"] = t(lambda: torch.bmm(xp.transpose(1, 2), wt.unsqueeze(0).expand(B, K, M)))
        r["pl->pl w"] = t(lambda: torch.bmm(w.unsqueeze(0).expand(B, M, K), xp))
        r["pl->pl wt"] = t(lambda: torch.bmm(wt.t().unsqueeze(0).expand(B, M, K), xp))
        print(f"L={L:5d} K={K:4d} M={M:4d}  " + "  ".join(f"{k} {v:6.1f}" for k, v in r.items()))


if __name__ == "__main__":
    main()
