"""Round 6 probe: weight gradient dW = gy^T x with bf16 operands and an fp32 result straight from the library (`torch.mm / addmm(..., out_dtype=torch.float32)`, aten::mm.dtype)
against the product's split-K batched GEMM + `phc_sum_slabs_bf16` (fast_ops.wgrad_split_k), per layer shape of the `im` learner.  Prints us per call (HIP events, 50 calls) and the error against fp64."""
import sys
import torch

sys.path.insert(0, ".")
from phc_amd.learning import fast_ops as fo  # noqa: E402

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


for B, N, K in ((16384, 1024, 1024), (16384, 512, 1024), (16384, 69, 512), (12288, 1024, 2048), (12288, 512, 1024), (4096, 1024, 2048)):
    gy = (torch.randn(B, N, device=dev) / B).to(torch.bfloat16)
    x = torch.randn(B, K, device=dev).to(torch.bfloat16)
    ref = gy.double().t() @ x.double()
    out = torch.zeros(N, K, device=dev)
    rows = [f"B {B:6d} N {N:5d} K {K:5d}"]
    fo.wgrad_split_k(gy, x, out=out)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    rows.append(f"split-K slabs + sum {t_us(lambda: fo.wgrad_split_k(gy, x, out=out)):7.1f} us (err {err:.1e})")
    gyt = gy.t()
    for name, fn in (("mm out_dtype=f32", lambda: torch.mm(gyt, x, out_dtype=torch.float32)),
                     ("addmm(out) out_dtype=f32", lambda: torch.addmm(out, gyt, x, out_dtype=torch.float32)),
                     ("mm bf16 out", lambda: torch.mm(gyt, x))):
        try:
            r = fn()
            err = float((r.double() - ref).abs().max() / ref.abs().max()) if "addmm" not in name else float("nan")
            rows.append(f"{name} {t_us(fn):7.1f} us (err {err:.1e})")
        except Exception as exc:   # noqa: BLE001
            rows.append(f"{name}: {type(exc).__name__} {str(exc)[:80]}")
    try:
        o2 = torch.zeros(N, K, device=dev)
        torch.mm(gyt, x, out_dtype=torch.float32, out=o2)
        rows.append(f"mm(out=) {t_us(lambda: torch.mm(gyt, x, out_dtype=torch.float32, out=o2)):7.1f} us")
    except Exception as exc:   # noqa: BLE001
        rows.append(f"mm(out=): {type(exc).__name__} {str(exc)[:80]}")
    print(" | ".join(rows), flush=True)
