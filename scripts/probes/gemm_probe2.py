"""dev tool: wgrad of the PPO layers -- one GEMM (what autograd issues) vs split-K as a batched GEMM + reduction; column sums"""
import time, torch
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
bf = torch.bfloat16
for B, K, N in ((16384, 934, 1024), (16384, 1024, 512), (16384, 512, 69), (12288, 1960, 1024), (12288, 1024, 512), (4096, 1960, 1024), (16384, 2048, 1536)):
    x = torch.randn(B, K, device="cuda", dtype=bf); dy = torch.randn(B, N, device="cuda", dtype=bf); w = torch.randn(N, K, device="cuda", dtype=bf)
    line = f"B={B} K={K} N={N}: fwd {t(lambda: torch.nn.functional.linear(x, w)):.1f}  dgrad {t(lambda: dy @ w):.1f}  wgrad {t(lambda: dy.t() @ x):.1f}"
    for C in (4, 8, 16, 32):
        f = lambda: torch.bmm(dy.view(C, B // C, N).transpose(1, 2), x.view(C, B // C, K)).float().sum(0)
        line += f"  splitK{C} {t(f):.1f}"
    ref = (dy.t().float() @ x.float())
    err = ((torch.bmm(dy.view(8, B // 8, N).transpose(1, 2), x.view(8, B // 8, K)).float().sum(0) - ref).abs().max() / ref.abs().max()).item()
    err0 = (((dy.t() @ x).float() - ref).abs().max() / ref.abs().max()).item()
    line += f"  colsum {t(lambda: dy.sum(0)):.1f} colsum_f32 {t(lambda: dy.sum(0, dtype=torch.float32)):.1f}  relerr splitK8 {err:.1e} direct {err0:.1e}"
    print(line, flush=True)
try:
    x = torch.randn(16384, 934, device="cuda", dtype=bf); dy = torch.randn(16384, 1024, device="cuda", dtype=bf)
    print("mm out_dtype fp32:", t(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32)))
except Exception as e:
    print("mm out_dtype unsupported:", type(e).__name__, str(e)[:100])
