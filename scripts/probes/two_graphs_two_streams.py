"""Do two LINEAR hipGraphs replayed on two streams run side by side?  Each graph: a chain of N dependent kernels of ~10 us on a small tensor (latency-bound,
a few workgroups: the GPU has room for both).  Prints wall time per iteration for: one graph alone, two graphs on two streams, and when the second graph starts
relative to the first (events)."""
import torch, time, sys
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
def chain(x):
    for _ in range(N):
        x = torch.tanh(x * 1.0001 + 0.1)
    return x
xa, xb = torch.randn(1 << 16, device=dev), torch.randn(1 << 16, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def capture(x, s):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): chain(x)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        y = chain(x)
    return g, y
ga, ya = capture(xa, sa)
gb, yb = capture(xb, sb)
main = torch.cuda.current_stream()
def it_one():
    sa.wait_stream(main)
    with torch.cuda.stream(sa): ga.replay()
    main.wait_stream(sa)
def it_two(ev=None):
    sa.wait_stream(main); sb.wait_stream(main)
    with torch.cuda.stream(sa):
        if ev: ev[0].record()
        ga.replay()
        if ev: ev[1].record()
    with torch.cuda.stream(sb):
        if ev: ev[2].record()
        gb.replay()
        if ev: ev[3].record()
    main.wait_stream(sa); main.wait_stream(sb)
def it_two_eager():
    sa.wait_stream(main); sb.wait_stream(main)
    with torch.cuda.stream(sa): chain(xa)
    with torch.cuda.stream(sb): chain(xb)
    main.wait_stream(sa); main.wait_stream(sb)
for name, f in (("one graph", it_one), ("two graphs, two streams", it_two), ("two eager chains, two streams", it_two_eager)):
    for _ in range(5): f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(30): f()
    th = time.perf_counter() - t
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t) / 30 * 1e3:.3f} ms per iteration (host {th / 30 * 1e3:.3f} ms)", flush=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
torch.cuda.synchronize()
it_two(ev)
torch.cuda.synchronize()
print("graph A: start 0, end %.3f ms; graph B: start %.3f, end %.3f ms" % (ev[0].elapsed_time(ev[1]), ev[0].elapsed_time(ev[2]), ev[0].elapsed_time(ev[3])))
