"""Does torch._addmm_activation fuse the ReLU into the hipBLASLt epilogue on this stack?  (round 2 probe)"""
import time
import torch

def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

for (M, K, N) in ((16384, 934, 1024), (16384, 1024, 512), (12288, 1960, 1024), (12288, 1024, 512), (4096, 934, 1024)):
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.03; b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    a = t(lambda: torch.addmm(b, x, w.t()))
    ar = t(lambda: torch.relu_(torch.addmm(b, x, w.t())))
    f = t(lambda: torch._addmm_activation(b, x, w.t()))
    y1 = torch.relu(torch.addmm(b, x, w.t())); y2 = torch._addmm_activation(b, x, w.t())
    print(f"M{M} K{K} N{N}: addmm {a:.1f} us, addmm+relu_ {ar:.1f} us, _addmm_activation {f:.1f} us, maxdiff {float((y1.float() - y2.float()).abs().max()):.3g}")
    gy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    print(f"   threshold_backward {t(lambda: torch.ops.aten.threshold_backward(gy, y1, 0.0)):.1f} us")
