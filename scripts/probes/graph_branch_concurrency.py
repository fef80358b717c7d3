"""Do independent branches of a captured hipGraph run concurrently on MI355X, and what is it worth for three PPO-sized MLPs?
Three independent bf16 MLP forward + backward chains (actor / critic / discriminator shapes of learning=im at minibatch 16384 / 3 x 4096 AMP rows), captured
(a) on one stream, (b) forked onto three streams inside the capture.  Prints ms per replay."""
import torch, time
dev = "cuda"
torch.manual_seed(0)
def mlp(i, hs, o):
    d, L = i, []
    for h in hs:
        L += [torch.nn.Linear(d, h), torch.nn.ReLU()]; d = h
    L.append(torch.nn.Linear(d, o))
    return torch.nn.Sequential(*L).to(dev).to(torch.bfloat16)
nets = [mlp(960, [1024, 512], 69), mlp(960, [1024, 512], 1), mlp(1984, [1024, 512], 1)]
xs = [torch.randn(16384, 960, device=dev, dtype=torch.bfloat16), torch.randn(16384, 960, device=dev, dtype=torch.bfloat16),
      torch.randn(12288, 1984, device=dev, dtype=torch.bfloat16)]
def body(streams):
    main = torch.cuda.current_stream()
    outs = []
    for n, x, s in zip(nets, xs, streams):
        if s is None:
            outs.append(n(x).float().square().mean())
        else:
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs.append(n(x).float().square().mean())
    for s in streams:
        if s is not None: main.wait_stream(s)
    torch.autograd.backward(outs)
def run(streams, tag):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): body(streams)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(streams)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50): g.replay()
    th = time.perf_counter() - t
    torch.cuda.synchronize()
    print(tag, "ms per replay", (time.perf_counter() - t) / 50 * 1e3, " host ms per replay() call", th / 50 * 1e3, flush=True)
    # eager too
    for _ in range(3): body(streams)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): body(streams)
    torch.cuda.synchronize()
    print(tag, "eager ms", (time.perf_counter() - t) / 20 * 1e3, flush=True)
run([None, None, None], "one stream   ")
run([None, torch.cuda.Stream(), torch.cuda.Stream()], "three streams")
run([None, None, None], "one stream   ")
