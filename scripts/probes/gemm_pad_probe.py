"""Does padding the first layer's K (obs 934 -> 960, AMP obs 1960 -> 1984 / 2048) speed the hipBLASLt GEMMs up?  (dev tool)"""
import time, torch
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / n * 1e6
dev, bf = "cuda", torch.bfloat16
for M, Ks, N in ((16384, (934, 936, 944, 960, 1024), 1024), (12288, (1960, 1984, 2048), 1024), (4096, (1960, 1984, 2048), 1024), (4096, (934, 960), 1024)):
    for K in Ks:
        x = torch.randn(M, K, device=dev, dtype=bf); w = torch.randn(N, K, device=dev, dtype=bf); b = torch.randn(N, device=dev, dtype=bf)
        gy = torch.randn(M, N, device=dev, dtype=bf)
        fwd = t(lambda: torch._addmm_activation(b, x, w.t()))
        gx = t(lambda: gy @ w)
        gw = t(lambda: torch.bmm(gy.view(8, M // 8, N).transpose(1, 2), x.view(8, M // 8, K)))
        fl = 2 * M * K * N / 1e12
        print(f"M {M:6d} K {K:5d} N {N}: fwd {fwd:6.1f} us ({fl/fwd*1e6:5.2f} PF/s)  gx {gx:6.1f} us  wgrad(split8) {gw:6.1f} us")
