"""dev tool: which piece of the optimizer step survives hipGraph capture at a given batch size (each piece in its own process)"""
import os, subprocess, sys
PIECES = ["linear", "fastlinear", "norm", "ppo", "disc", "adam", "stack"]
if len(sys.argv) > 2:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    piece, B = sys.argv[1], int(sys.argv[2])
    from phc_amd.learning.fast_ops import FastLinear, adam_clip_step, ppo_loss
    from phc_amd.learning.running_mean_std import RunningMeanStd
    dev = "cuda"
    torch.manual_seed(0)
    if piece in ("linear", "fastlinear"):
        L = (FastLinear if piece == "fastlinear" else torch.nn.Linear)
        net = torch.nn.Sequential(L(934, 1024), torch.nn.ReLU(), L(1024, 512), torch.nn.ReLU(), L(512, 69)).cuda()
        x = torch.randn(B, 934, device=dev).to(torch.bfloat16)
        def body():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = net(x)
            y.float().square().mean().backward()
    elif piece == "norm":
        m = RunningMeanStd(934).cuda().train()
        x = torch.randn(4 * B, 934, device=dev); idx = torch.randperm(4 * B, device=dev)[:B]
        def body():
            m(x, row_index=idx, out_dtype=torch.bfloat16)
    elif piece == "ppo":
        D = 69
        mu = torch.randn(B, D, device=dev).to(torch.bfloat16).requires_grad_(True); val = torch.randn(B, 1, device=dev).to(torch.bfloat16).requires_grad_(True)
        ls = torch.full((D,), -2.9, device=dev); a = torch.randn(B, D, device=dev); o = torch.randn(B, device=dev); r = torch.randn(B, 1, device=dev)
        sg = torch.exp(ls).expand(B, D).contiguous()
        def body():
            loss, st = ppo_loss(mu, val, ls, a, o, o, r, r, a, sg, 0.2, 5.0, 0.0, 10.0, False, unit_grad=True)
            loss.backward()
    elif piece == "disc":
        net = torch.nn.Sequential(torch.nn.Linear(1960, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 512), torch.nn.ReLU(), torch.nn.Linear(512, 1)).cuda()
        m = max(B // 4, 8)
        xa = torch.randn(2 * m, 1960, device=dev).to(torch.bfloat16); xd = torch.randn(m, 1960, device=dev).to(torch.bfloat16)
        def body():
            d = xd.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                lg = net(torch.cat([xa, d], 0)).float()
            la, ld = lg[:2 * m], lg[2 * m:]
            bce = torch.nn.BCEWithLogitsLoss()
            loss = 0.5 * (bce(la, torch.zeros_like(la)) + bce(ld, torch.ones_like(ld)))
            g = torch.autograd.grad(ld, d, grad_outputs=torch.ones_like(ld), create_graph=True, retain_graph=True)[0].float()
            (loss + 5 * g.square().sum(-1).mean()).backward()
    elif piece == "adam":
        p = torch.randn(5_000_000, device=dev, requires_grad=True); p.grad = torch.randn_like(p)
        opt = torch.optim.Adam([p], 1e-3)
        adam_clip_step(opt, p, p.grad, 50.0)
        stp = torch.zeros((), dtype=torch.int64, device=dev); sh = torch.zeros(5_000_000, dtype=torch.bfloat16, device=dev)
        def body():
            adam_clip_step(opt, p, p.grad, 50.0, shadow=sh, step_device=stp, count_host=False)
    elif piece == "stack":
        acc = torch.zeros(4, device=dev); vals = [torch.randn((), device=dev) for _ in range(4)]
        def body():
            acc.add_(torch.stack([v.float().reshape(()) for v in vals]))
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): body()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay(); torch.cuda.synchronize()
    print("ok")
else:
    B = sys.argv[1] if len(sys.argv) > 1 else "2048"
    for p in PIECES:
        r = subprocess.run([sys.executable, __file__, p, B], capture_output=True, text=True)
        print(f"B={B} {p:11s} rc={r.returncode} {r.stdout.strip()[-20:]} {r.stderr.strip().splitlines()[-1][:120] if r.returncode and r.stderr.strip() else ''}", flush=True)
