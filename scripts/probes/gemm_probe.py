"""dev tool: bf16 GEMM timings for the PPO shapes (forward, dgrad, wgrad) with and without K/N padding"""
import time, torch
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
B = 16384
for K in (934, 936, 944, 960, 1960, 1984):
    x = torch.randn(B, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(1024, K, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(B, 1024, device="cuda", dtype=torch.bfloat16)
    print(K, "fwd %.1f us" % t(lambda: torch.nn.functional.linear(x, w)), "wgrad %.1f us" % t(lambda: dy.t() @ x), "dgrad %.1f us" % t(lambda: dy @ w))
