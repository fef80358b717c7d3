"""Round 6 probe: actor and critic trunks of `learning=im` have identical shapes (934 -> 1024 -> 512) and independent weights -- would ONE batched GEMM (batch 2) per layer and direction beat the two
separate GEMMs the policy pass issues (512 instead of 256 output tiles per launch on 256 CUs)?  us per call, HIP events, 50 calls; bf16."""
import torch

dev = "cuda"
torch.manual_seed(0)


def t_us(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = 16384
tot1 = tot2 = 0.0
for name, (M, K, N) in (("fwd L1  x[B,1024p] W[1024,1024p]^T", (B, 1024, 1024)), ("fwd L2  x[B,1024] W[512,1024]^T", (B, 1024, 512)),
                        ("dX  L2  g[B,512] W[512,1024]", (B, 512, 1024)), ("dW  L1  (split-K 8) g^T x", (1024, B // 8, 1024)), ("dW  L2  (split-K 8) g^T x", (512, B // 8, 1024))):
    if name.startswith("dW"):
        a1, b1 = torch.randn(8, M, K, device=dev).bfloat16(), torch.randn(8, K, N, device=dev).bfloat16()
        a2, b2 = torch.randn(16, M, K, device=dev).bfloat16(), torch.randn(16, K, N, device=dev).bfloat16()
        one, two = t_us(lambda: torch.bmm(a1, b1)), t_us(lambda: torch.bmm(a2, b2))
    else:
        a1, b1 = torch.randn(M, K, device=dev).bfloat16(), torch.randn(K, N, device=dev).bfloat16()
        a2, b2 = torch.randn(2, M, K, device=dev).bfloat16(), torch.randn(2, K, N, device=dev).bfloat16()
        one, two = t_us(lambda: torch.mm(a1, b1)), t_us(lambda: torch.bmm(a2, b2))
    tot1 += 2 * one
    tot2 += two
    print(f"{name:40s} one net {one:6.1f} us  x2 = {2 * one:6.1f} us   both nets in one batched launch {two:6.1f} us  ({two / (2 * one):.2f} x)", flush=True)
print(f"sum over the five products: two launches each {tot1:.1f} us, one batched launch each {tot2:.1f} us")
