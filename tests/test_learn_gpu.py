"""Learner-side HIP kernels against the torch expressions that define them (which test_learner_cpu.py pins to the reference)."""
import copy

import numpy as np
import pytest
import torch

from phc_amd.learning.running_mean_std import RunningMeanStd

pytestmark = pytest.mark.gpu


def _pair(cols, seed=0):
    g = torch.Generator().manual_seed(seed)
    cpu = RunningMeanStd(cols)
    cpu.running_mean.copy_(torch.randn(cols, generator=g, dtype=torch.float64) * 0.5)
    cpu.running_var.copy_(torch.rand(cols, generator=g, dtype=torch.float64) * 2 + 0.05)
    cpu.count.fill_(1234.0)
    dev = copy.deepcopy(cpu).cuda()
    return cpu, dev, g


@pytest.mark.parametrize("rows,cols", [(5000, 934), (4096, 1960), (70, 3), (2, 934)])
def test_running_norm_kernel_equals_torch_path(rows, cols):
    """phc_running_norm == RunningMeanStd.forward (running_mean_std.py:69-111): normalised + clamped output bit-equal to the IEEE
    fp32 evaluation of the reference expression (numpy; torch's AVX-512 CPU kernels deviate from it by up to 2 ulp themselves), bf16
    output == its rounding, running statistics after three train-mode batches to fp64 round-off of the fp32 batch moments."""
    cpu, dev, g = _pair(cols)
    cpu.train(); dev.train()
    for it in range(3):
        x = torch.randn(rows, cols, generator=g) * (1 + it) + 0.3 * it
        x[0, 0] = 40.0   # clamped
        m32, v32 = dev.running_mean.cpu().numpy().astype(np.float32), dev.running_var.cpu().numpy().astype(np.float32)   # the kernel's own statistics
        exact = np.clip((x.numpy() - m32) / np.sqrt(v32 + np.float32(1e-5)), -5.0, 5.0)
        y_ref = cpu(x)
        y = dev(x.cuda())
        np.testing.assert_array_equal(y.cpu().numpy(), exact)
        torch.testing.assert_close(y.cpu(), y_ref, rtol=3e-7, atol=1e-6)
        if rows > 1:
            np.testing.assert_allclose(dev.running_mean.cpu().numpy(), cpu.running_mean.numpy(), rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(dev.running_var.cpu().numpy(), cpu.running_var.numpy(), rtol=2e-5)
        assert float(dev.count) == float(cpu.count)
    # eval mode: statistics untouched, bf16 output == rounding of the fp32 output
    cpu.eval(); dev.eval()
    before = dev.running_mean.clone()
    x = torch.randn(rows, cols, generator=g)
    yb = dev(x.cuda(), out_dtype=torch.bfloat16)
    assert yb.dtype == torch.bfloat16 and torch.equal(yb.cpu(), dev(x.cuda()).cpu().to(torch.bfloat16)) and torch.equal(before, dev.running_mean)


def test_running_norm_frozen_source_and_update_only():
    """`norm_from` (amp_agent.py:527-532: the output comes from the frozen epoch-start copy while the live statistics update) and
    the store-free update (`want_output=False`)."""
    cpu, dev, g = _pair(358, seed=3)
    frozen_cpu, frozen_dev = copy.deepcopy(cpu), copy.deepcopy(dev)
    frozen_cpu.freeze(); frozen_dev.freeze()
    cpu.train(); dev.train()
    x = torch.randn(3000, 358, generator=g) * 3
    out_ref = frozen_cpu(x)
    cpu(x)
    out = dev(x.cuda(), norm_from=frozen_dev)
    torch.testing.assert_close(out.cpu(), out_ref, rtol=3e-7, atol=1e-6)
    np.testing.assert_allclose(dev.running_mean.cpu().numpy(), cpu.running_mean.numpy(), rtol=1e-6, atol=1e-7)
    assert torch.equal(frozen_dev.running_mean.cpu(), frozen_cpu.running_mean)
    cpu(x)
    assert dev(x.cuda(), want_output=False) is None
    np.testing.assert_allclose(dev.running_var.cpu().numpy(), cpu.running_var.numpy(), rtol=2e-5)
    x[5, 7] = float("nan")
    dev.eval()
    assert torch.isnan(dev(x.cuda())[5, 7])


@pytest.mark.parametrize("rows,cols", [(16384, 1024), (16384, 69), (4096, 1), (1000, 513)])
def test_colsum_kernel(rows, cols):
    from phc_amd.learning.fast_ops import colsum_bf16
    x = (torch.randn(rows, cols, device="cuda") * 0.1).to(torch.bfloat16)
    ref = x.double().sum(0)
    got = colsum_bf16(x)
    assert got.dtype == torch.float32
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-4)


def test_deferred_column_sums_finish_in_one_batch_launch():
    """Round 5: inside `deferred_colsums()` the column sums that go straight into a destination only run their first stage; one
    `phc_colsum_finish_batch` launch at the end finishes all of them (20 jobs here: two launches of <= 16) -- same values as the immediate path,
    incl. the masked variant, the value head's [cols + 1] result and an accumulating job."""
    import ctypes as C
    from phc_amd import _lib as L
    from phc_amd.learning import fast_ops as fo
    torch.manual_seed(0)
    shapes = [(16384, 1024), (16384, 512), (4096, 69), (1000, 513)] * 5
    xs = [(torch.randn(r, c, device="cuda") * 0.1).to(torch.bfloat16) for r, c in shapes]
    ys = [torch.randn(r, c, device="cuda").to(torch.bfloat16) for r, c in shapes]
    want = [fo.colsum_relu_bf16(x, y) if i % 2 else (None, fo.colsum_bf16(x)) for i, (x, y) in enumerate(zip(xs, ys))]
    outs = [torch.full((c,), 7.0, device="cuda") for _, c in shapes]
    with fo.deferred_colsums():
        got = [fo.colsum_relu_bf16(x, y, out=o) if i % 2 else (None, fo.colsum_bf16(x, out=o)) for i, (x, y, o) in enumerate(zip(xs, ys, outs))]
        assert sum(len(v) for v in fo._pending.values()) == len(shapes)
    assert not fo._pending
    torch.cuda.synchronize()
    for i, ((gm_w, w), (gm_g, g), o) in enumerate(zip(want, got, outs)):
        assert g is o
        torch.testing.assert_close(o, w, rtol=0, atol=0)
        if gm_w is not None:
            assert torch.equal(gm_w, gm_g)
    # an accumulating job through the C ABI itself
    lib = L.load()
    x = xs[0]
    ws = torch.empty(lib.phc_colsum_workspace(*x.shape) // 4, device="cuda")
    L.check(lib.phc_colsum_bf16(x.data_ptr(), x.shape[0], x.shape[1], None, ws.data_ptr(), torch.cuda.current_stream().cuda_stream), "first stage")
    acc = torch.full((x.shape[1],), 2.0, device="cuda")
    job = (L.ColsumJob * 1)()
    job[0].partial, job[0].out, job[0].nchunks, job[0].cols, job[0].accumulate = ws.data_ptr(), acc.data_ptr(), lib.phc_colsum_chunks(x.shape[0]), x.shape[1], 1
    L.check(lib.phc_colsum_finish_batch(1, job, torch.cuda.current_stream().cuda_stream), "batch")
    torch.testing.assert_close(acc, want[0][1] + 2.0, rtol=1e-6, atol=1e-6)
    assert lib.phc_colsum_finish_batch(1, None, None) != 0      # PHC_EINVAL


def test_a_layer_applied_twice_inside_deferred_colsums_keeps_both_bias_contributions():
    """inside `deferred_colsums()` a bucket parameter's FIRST bias gradient is a pending job whose finishing launch STORES at the end of the pass; a second
    application of the same module in that pass adds its contribution right away -- the store must land first.  One FastLinear + ReLU, one plain FastLinear and one
    one-output layer (the value head's kernels), each applied to two different inputs in ONE backward pass: gradients == two separate passes summed."""
    from phc_amd.learning.amp_agent import FlatGradBucket
    from phc_amd.learning import fast_ops as fo
    from phc_amd.learning.network import build_mlp
    torch.manual_seed(3)
    B, K, H = 4096, 96, 64
    trunk = build_mlp(K, [H], "relu", fo.FastLinear).cuda()
    mid, head = fo.FastLinear(H, H).cuda(), fo.FastLinear(H, 1).cuda()
    params = [p for m in (trunk, mid, head) for p in m.parameters()]
    bucket = FlatGradBucket(params)
    xa, xb = torch.randn(B, K, device="cuda"), torch.randn(B, K, device="cuda") * 0.5 + 0.3

    def loss_of(x):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return (head(mid(trunk(x))).float() ** 2).mean()

    singles = []
    with bucket.shadow_scope():
        for x in (xa, xb):
            bucket.zero()
            with fo.deferred_colsums():
                loss_of(x).backward()
            singles.append(bucket.flat.clone())
        bucket.zero()
        with fo.deferred_colsums():
            (loss_of(xa) + loss_of(xb)).backward()      # every module twice in one pass
        assert not fo._pending and not fo._pending_dst
    both = bucket.flat.clone()
    torch.cuda.synchronize()
    want = singles[0] + singles[1]
    for p, (o, k) in zip(bucket.params, bucket.segments):
        scale = float(want[o:o + k].abs().max())
        assert scale > 0 and float((both[o:o + k] - want[o:o + k]).abs().max()) <= 2e-2 * scale, (tuple(p.shape), scale)
    for m in (trunk[0], mid, head):      # the bias gradients in particular (the store that used to come last)
        o, k = bucket.segments[[id(q) for q in bucket.params].index(id(m.bias))]
        torch.testing.assert_close(both[o:o + k], want[o:o + k], rtol=2e-2, atol=2e-2 * float(want[o:o + k].abs().max()))


@pytest.mark.parametrize("B,K,N", [(16384, 934, 1024), (16384, 512, 69), (1000, 130, 7)])
def test_fast_linear_matches_autocast_linear(B, K, N):
    """FastLinear's training pass == nn.Linear under bf16 autocast: same forward values; weight / bias / input gradients equal to the
    fp32 gradients within bf16 GEMM accuracy, and at least as close to them as autograd's own bf16 path."""
    from phc_amd.learning.fast_ops import FastLinear
    torch.manual_seed(0)
    ref = torch.nn.Linear(K, N).cuda()
    fast = FastLinear(K, N).cuda()
    fast.load_state_dict(ref.state_dict())
    x = torch.randn(B, K, device="cuda")
    gy = torch.randn(B, N, device="cuda") / B
    outs = {}
    for name, mod in (("ref", ref), ("fast", fast)):
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = mod(xi)
        assert y.dtype == torch.bfloat16
        (y.float() * gy).sum().backward()
        outs[name] = (y.float(), mod.weight.grad.clone(), mod.bias.grad.clone(), xi.grad.clone())
    assert torch.equal(outs["ref"][0], outs["fast"][0])
    x64 = x.double().requires_grad_(True)
    y64 = x64 @ ref.weight.double().t() + ref.bias.double()
    (y64 * gy.double()).sum().backward(inputs=[x64])
    gw64, gb64 = gy.double().t() @ x.double(), gy.double().sum(0)
    for k, exact in ((1, gw64), (2, gb64), (3, x64.grad)):
        scale = exact.abs().max().item()
        e_fast = (outs["fast"][k].double() - exact).abs().max().item() / scale
        e_ref = (outs["ref"][k].double() - exact).abs().max().item() / scale
        assert e_fast < 2e-2 and e_fast <= 1.5 * e_ref + 1e-4, (k, e_fast, e_ref)
    # no_grad / frozen / fp32 paths are plain nn.Linear
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        assert torch.equal(fast(x), ref(x))
    assert torch.equal(fast(x), ref(x))


def test_pnn_lateral_path_on_the_bf16_device_passes():
    """pnn.py:103: the lateral term enters BEFORE the second hidden layer's activation, relu(W a1 + b + lateral).  On the device passes that
    layer's ReLU normally rides in the GEMM epilogue (FastLinear.fuse_relu) -- it must not for a lateral column: the bf16 autocast
    training forward / backward and the no-grad inference forward equal a plain nn.Linear / nn.ReLU restatement of the reference's forward."""
    from phc_amd.learning.network import PNN
    torch.manual_seed(3)
    K, A, B = 160, 12, 512
    pnn = PNN(K, [64, 32], "relu", A, 3, has_lateral=True).cuda()
    x = torch.randn(B, K, device="cuda")

    def plain(xi):   # the reference's forward on plain ops with the same parameters (pnn.py:85-126)
        first, outs = [], []
        for k in range(3):
            col = pnn.actors[k]
            a1 = torch.relu(torch.nn.functional.linear(xi, col[0].weight, col[0].bias))
            lat = sum(torch.nn.functional.linear(first[j], pnn.u[k - 1][j][0].weight) for j in range(len(first))) if first else 0
            a2 = torch.relu(torch.nn.functional.linear(a1, col[2].weight, col[2].bias) + lat)
            outs.append(torch.nn.functional.linear(a2, col[4].weight, col[4].bias))
            first.append(a1)
        return outs
    res = {}
    for name in ("plain", "pnn"):
        pnn.zero_grad()
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            outs = plain(xi) if name == "plain" else pnn(xi, idx=-1)[1]
        y = torch.stack([o.float() for o in outs], 1)
        y.square().mean().backward()
        res[name] = (y.detach(), xi.grad.clone(), pnn.actors[2][2].weight.grad.clone(), pnn.u[1][0][0].weight.grad.clone())
    assert float((res["plain"][0] - res["pnn"][0]).abs().max()) <= 2e-2 * float(res["plain"][0].abs().max())
    # the un-fixed path (ReLU before the lateral term) differs from the reference by O(1) of the output scale on the lateral columns
    for k in (1, 2, 3):
        scale = float(res["plain"][k].abs().max())
        assert float((res["plain"][k] - res["pnn"][k]).abs().max()) < 3e-2 * scale, k
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y_inf = torch.stack([o.float() for o in pnn(x, idx=-1)[1]], 1)
    assert float((y_inf - res["plain"][0]).abs().max()) <= 2e-2 * float(res["plain"][0].abs().max())
    assert not any(col[2].fuse_relu for col in pnn.actors) and all(col[0].fuse_relu for col in pnn.actors)


@pytest.mark.parametrize("B,K,N", [(16384, 934, 1024), (4096, 1024, 512), (1000, 130, 7)])
def test_fused_relu_layer_matches_linear_plus_relu(B, K, N):
    """build_mlp's FastLinear + FusedReLU pair (ReLU in the GEMM epilogue, ReLU mask + bias gradient in one pass, phc_colsum_relu_bf16) ==
    nn.Linear + nn.ReLU under bf16 autocast: identical forward values, identical masked gradients up to the bf16 GEMMs' accuracy; the
    rollout-inference path (bf16 shadow parameters, no grad) fuses too; state-dict keys are those of Linear, ReLU."""
    from phc_amd.learning.network import build_mlp
    from phc_amd.learning.fast_ops import FastLinear, FusedReLU
    torch.manual_seed(1)
    ref = torch.nn.Sequential(torch.nn.Linear(K, N), torch.nn.ReLU()).cuda()
    fast = build_mlp(K, [N], "relu", FastLinear).cuda()
    assert isinstance(fast[0], FastLinear) and fast[0].fuse_relu and isinstance(fast[1], FusedReLU) and list(fast.state_dict()) == list(ref.state_dict())
    fast.load_state_dict(ref.state_dict())
    x = torch.randn(B, K, device="cuda")
    gy = torch.randn(B, N, device="cuda") / B
    outs = {}
    for name, mod in (("ref", ref), ("fast", fast)):
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = mod(xi)
        (y.float() * gy).sum().backward()
        outs[name] = (y.float(), mod[0].weight.grad.clone(), mod[0].bias.grad.clone(), xi.grad.clone())
    assert torch.equal(outs["ref"][0], outs["fast"][0]) and float((outs["fast"][0] == 0).float().mean()) > 0.2
    for k in (1, 2, 3):
        scale = outs["ref"][k].abs().max().item()
        assert (outs["fast"][k] - outs["ref"][k]).abs().max().item() < 2e-2 * scale, k
    # bias gradient: exactly the column sums of the masked bf16 output gradient
    gm = torch.where(outs["ref"][0] > 0, gy.to(torch.bfloat16).float(), torch.zeros_like(gy))
    np.testing.assert_allclose(outs["fast"][2].cpu().numpy(), gm.sum(0).cpu().numpy(), rtol=2e-3, atol=1e-6)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        assert torch.equal(fast(x), ref(x))
    assert torch.equal(fast(x), ref(x))          # fp32, no autocast: plain modules


def test_adam_clip_step_equals_torch():
    """phc_adam_clip_step == clip_grad_norm_ + torch.optim.Adam.step on the flat parameter (three steps, clipping active and
    inactive, weight decay on), and the optimizer's state_dict stays loadable by a plain torch Adam."""
    from phc_amd.learning.fast_ops import adam_clip_step
    torch.manual_seed(1)
    n = 1_000_003
    p0 = torch.randn(n, device="cuda")
    pa, pb = p0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
    oa = torch.optim.Adam([pa], 3e-3, eps=1e-8, weight_decay=1e-3)
    ob = torch.optim.Adam([pb], 3e-3, eps=1e-8, weight_decay=1e-3)
    pb.grad = torch.zeros_like(pb)
    shadow = torch.zeros(n, device="cuda", dtype=torch.bfloat16)
    for it, gscale in enumerate((1.0, 1e-3, 0.05)):
        g = torch.randn(n, device="cuda") * gscale
        pa.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([pa], 50.0)
        oa.step()
        pb.grad.copy_(g)
        adam_clip_step(ob, pb, pb.grad, 50.0, shadow=shadow)
        assert torch.equal(shadow, pb.detach().to(torch.bfloat16))
        torch.testing.assert_close(pb.grad, pa.grad, rtol=5e-5, atol=1e-9)   # clip coefficient: fp64 sum of squares here, fp32 norm in torch
        torch.testing.assert_close(pb.detach(), pa.detach(), rtol=5e-5, atol=2e-6)
    sa, sb = oa.state[pa], ob.state[pb]
    assert float(sb["step"]) == 3.0 == float(sa["step"])
    torch.testing.assert_close(sb["exp_avg_sq"], sa["exp_avg_sq"], rtol=1e-4, atol=1e-12)   # squares of the clipped gradient: twice its relative difference
    oc = torch.optim.Adam([p0.clone().requires_grad_(True)], 3e-3)
    oc.load_state_dict(ob.state_dict())


@pytest.mark.parametrize("dtype,clip_value,D", [(torch.bfloat16, False, 69), (torch.float32, True, 37), (torch.float32, False, 153)])
def test_fused_ppo_loss_equals_torch_losses(dtype, clip_value, D):
    """phc_ppo_loss == the torch expressions of IMAmpAgent._ppo_loss_torch (amp_agent.py:598-640): loss, the five statistics and the
    gradients w.r.t. the mu and value heads (autograd of the torch path), incl. rows outside the clip range and actions beyond +-1."""
    from phc_amd.learning.fast_ops import ppo_loss
    from phc_amd.learning.network import ModelAMPContinuous, policy_kl
    torch.manual_seed(4)
    B, dev = 5000, "cuda"
    e_clip, cc, ec, bl = 0.2, 5.0, 0.01, 10.0
    logstd = torch.full((D,), -2.9, device=dev) + torch.randn(D, device=dev) * 0.1
    mu0 = (torch.randn(B, D, device=dev) * 0.7).to(dtype)
    val0 = torch.randn(B, 1, device=dev).to(dtype)
    old_mu = mu0.float() + torch.randn(B, D, device=dev) * 0.02
    old_sigma = torch.exp(logstd).expand(B, D).contiguous()
    actions = old_mu + old_sigma * torch.randn(B, D, device=dev)
    old_nlp = ModelAMPContinuous.neglogp(actions, old_mu, old_sigma, logstd.expand(B, D))
    adv = torch.randn(B, device=dev)
    ret = torch.randn(B, 1, device=dev)
    old_val = val0.float() + torch.randn(B, 1, device=dev) * 0.3

    mu_t, val_t = mu0.clone().requires_grad_(True), val0.clone().requires_grad_(True)
    mu, value = mu_t.float(), val_t.float()
    sigma = torch.exp(logstd).expand(B, D)
    nlp = ModelAMPContinuous.neglogp(actions, mu, sigma, logstd.expand(B, D))
    ratio = torch.exp(old_nlp - nlp)
    a_loss = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1 - e_clip, 1 + e_clip)).mean()
    if clip_value:
        vpc = old_val + (value - old_val).clamp(-e_clip, e_clip)
        c_loss = torch.max((value - ret) ** 2, (vpc - ret) ** 2).mean()
    else:
        c_loss = ((ret - value) ** 2).mean()
    b_loss = ((torch.clamp_min(mu - 1, 0) ** 2) + (torch.clamp_max(mu + 1, 0) ** 2)).sum(-1).mean()
    ent = (0.5 + 0.5 * np.log(2 * np.pi) + logstd).sum()
    ref = a_loss + cc * c_loss - ec * ent + bl * b_loss
    ref.backward()
    kl = policy_kl(mu.detach(), sigma, old_mu, old_sigma)
    assert ((ratio < 1 - e_clip) | (ratio > 1 + e_clip)).float().mean() > 0.05 and (mu0.float().abs() > 1).any()

    mu_f, val_f = mu0.clone().requires_grad_(True), val0.clone().requires_grad_(True)
    loss, st = ppo_loss(mu_f, val_f, logstd, actions, old_nlp, adv, ret, old_val, old_mu, old_sigma, e_clip, cc, ec, bl, clip_value, unit_grad=True)
    (loss * 1.0).backward()
    tol = dict(rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(loss, ref.detach(), **tol)
    clip_frac = ((ratio - 1.0).abs() > e_clip).float().mean()       # `actor_clipped` (common_agent.py:570-571)
    torch.testing.assert_close(st[:5], torch.stack([a_loss, c_loss, b_loss, ent, kl]).detach(), **tol)
    assert abs(float(st[5]) - float(clip_frac)) <= 3.0 / B and float(clip_frac) > 0.05      # (a count: a row whose ratio sits on the threshold may fall on either side)
    gtol = dict(rtol=2e-2, atol=2e-6) if dtype == torch.bfloat16 else dict(rtol=2e-4, atol=2e-7)
    torch.testing.assert_close(mu_f.grad.float(), mu_t.grad.float(), **gtol)
    torch.testing.assert_close(val_f.grad.float(), val_t.grad.float(), **gtol)
    # general incoming gradient
    mu_g, val_g = mu0.clone().requires_grad_(True), val0.clone().requires_grad_(True)
    loss2, _ = ppo_loss(mu_g, val_g, logstd, actions, old_nlp, adv, ret, old_val, old_mu, old_sigma, e_clip, cc, ec, bl, clip_value)
    (3.0 * loss2).backward()
    torch.testing.assert_close(mu_g.grad.float(), 3.0 * mu_t.grad.float(), rtol=3e-2 if dtype == torch.bfloat16 else 2e-4, atol=1e-5)


def test_row_index_variants_equal_gathered_inputs():
    """phc_running_norm / phc_ppo_loss with a row index == the same kernels on the gathered minibatch (bit-identical: same arithmetic,
    rows read in place)."""
    from phc_amd.learning.fast_ops import ppo_loss
    torch.manual_seed(7)
    dev = "cuda"
    N, B, C, D = 20000, 4096, 358, 69
    x = torch.randn(N, C, device=dev)
    idx = torch.randperm(N, device=dev)[:B]
    a, b = RunningMeanStd(C).cuda().train(), RunningMeanStd(C).cuda().train()
    ya, yb = a(x, row_index=idx, out_dtype=torch.bfloat16), b(x[idx].contiguous(), out_dtype=torch.bfloat16)
    assert torch.equal(ya, yb) and torch.equal(a.running_mean, b.running_mean) and torch.equal(a.running_var, b.running_var) and float(a.count) == float(b.count)
    logstd = torch.full((D,), -2.9, device=dev)
    mu, val = (torch.randn(B, D, device=dev) * 0.5).to(torch.bfloat16), torch.randn(B, 1, device=dev).to(torch.bfloat16)
    act, omu = torch.randn(N, D, device=dev) * 0.5, torch.randn(N, D, device=dev) * 0.5
    osig = torch.exp(logstd).expand(N, D).contiguous()
    onlp, adv, ret, oval = torch.randn(N, device=dev) * 3 + 60, torch.randn(N, device=dev), torch.randn(N, 1, device=dev), torch.randn(N, 1, device=dev)
    outs = []
    for ri in (idx, None):
        g = (lambda t: t) if ri is not None else (lambda t: t[idx].contiguous())
        m, v = mu.clone().requires_grad_(True), val.clone().requires_grad_(True)
        loss, st = ppo_loss(m, v, logstd, g(act), g(onlp), g(adv), g(ret), g(oval), g(omu), g(osig), 0.2, 5.0, 0.0, 10.0, True, unit_grad=True, row_index=ri)
        loss.backward()
        outs.append((loss.detach(), st, m.grad, v.grad))
    for p, q in zip(*outs):
        assert torch.equal(p, q)


@pytest.mark.parametrize("mode", ["plain", "fused_relu", "agent_path"])
def test_twice_differentiable_linear_matches_autograd(mode):
    """FastLinearDD (discriminator MLP): loss + gradient penalty -- which differentiates the backward pass -- give the same parameter
    gradients as nn.Linear under bf16 autocast, within bf16 GEMM accuracy of the fp64 result.  `fused_relu`: the pair FastLinearDD +
    FusedReLU as network.build_mlp makes it (ReLU in the GEMM epilogue, its mask inside the twice-differentiable backward node).
    `agent_path`: additionally everything IMAmpAgent._fwd_bwd does around it -- the input is one buffer whose last row block is the
    leaf (rows_with_grad), the logit layer is FastLinear1DD, the penalty's cotangent is a row mask over ALL logits inside
    input_grad_only(row_start) (layers work on the demo rows alone and leave the other rows of their gradients unwritten), and the
    loss backward runs inside param_grad_only()."""
    from phc_amd.learning.fast_ops import FastLinear1DD, FastLinearDD, input_grad_only, param_grad_only, rows_with_grad
    from phc_amd.learning.network import build_mlp
    torch.manual_seed(3)
    B, m, K = 6144, 2048, 1960
    fused_relu, agent_path = mode != "plain", mode == "agent_path"

    def build(linear, dtype=torch.float32):
        if fused_relu and linear is FastLinearDD:
            net = build_mlp(K, [1024, 512], "relu", FastLinearDD)
            assert net[0].fuse_relu and net[2].fuse_relu
            net.append(FastLinear1DD(512, 1) if agent_path else torch.nn.Linear(512, 1))
            return net.cuda().to(dtype)
        net = torch.nn.Sequential(linear(K, 1024), torch.nn.ReLU(), linear(1024, 512), torch.nn.ReLU(), torch.nn.Linear(512, 1)).cuda().to(dtype)
        return net
    ref = build(torch.nn.Linear)
    fast = build(FastLinearDD)
    fast.load_state_dict(ref.state_dict())
    ref64 = build(torch.nn.Linear, torch.float64)
    ref64.load_state_dict(ref.state_dict())
    xa = (torch.randn(B - m, K, device="cuda") * 0.5).to(torch.bfloat16)
    xd = (torch.randn(m, K, device="cuda") * 0.5).to(torch.bfloat16)

    def run(net, f64=False, agent=False):
        bce = torch.nn.BCEWithLogitsLoss()
        if agent:
            buf = torch.cat([xa, xd], 0)
            d = buf[B - m:].requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                raw = net(rows_with_grad(buf, d, B - m))
            assert raw.dtype == torch.bfloat16
            lg = raw.float()
            mask = torch.zeros((B, 1), dtype=raw.dtype, device="cuda")
            mask[B - m:] = 1
            with input_grad_only(row_start=B - m):
                g = torch.autograd.grad(raw, d, grad_outputs=mask, create_graph=True, retain_graph=True)[0].float()
        else:
            d = (xd.double() if f64 else xd.clone()).requires_grad_(True)
            a = xa.double() if f64 else xa
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not f64):
                lg = net(torch.cat([a, d], 0))
            lg = lg if f64 else lg.float()
            with input_grad_only():
                g = torch.autograd.grad(lg[B - m:], d, grad_outputs=torch.ones_like(lg[B - m:]), create_graph=True, retain_graph=True)[0]
            g = g if f64 else g.float()
        la, ld = lg[:B - m], lg[B - m:]
        loss = 0.5 * (bce(la, torch.zeros_like(la)) + bce(ld, torch.ones_like(ld)))
        pen = g.square().sum(-1).mean()
        if agent:
            with param_grad_only():
                (loss + 5.0 * pen).backward()
            assert d.grad is None      # the input gradient of the loss is not formed
        else:
            (loss + 5.0 * pen).backward()
        return float(loss), float(pen), [p.grad.double().clone() for p in net.parameters()]
    l64, p64, g64 = run(ref64, True)
    lr_, pr, gr = run(ref)
    lf, pf, gf = run(fast, agent=agent_path)
    assert abs(lf - lr_) < (2e-4 if agent_path else 1e-6) and abs(pf - pr) <= 2e-3 * abs(pr) + 1e-9, (lf, lr_, pf, pr)   # (agent_path: bf16 logits from the one-output kernel)
    for k, (a, b, e) in enumerate(zip(gf, gr, g64)):
        scale = e.abs().max().item() + 1e-12
        ef, er = (a - e).abs().max().item() / scale, (b - e).abs().max().item() / scale
        assert ef < 0.1 and ef <= 1.5 * er + 2e-3, (k, ef, er)   # er: what stock autograd's bf16 path achieves (~5 % on the small bias gradients)


@pytest.mark.parametrize("B,K", [(5000, 512), (16384, 512), (4099, 1024), (777, 130)])
def test_one_output_linear_kernels(B, K):
    """FastLinear with out_features == 1 (the value head): phc_linear1_forward / _backward == nn.Linear under bf16 autocast (ragged last blocks: B = 5000,
    4099, 777)."""
    from phc_amd.learning.fast_ops import FastLinear
    torch.manual_seed(5)
    ref, fast = torch.nn.Linear(K, 1).cuda(), FastLinear(K, 1).cuda()
    fast.load_state_dict(ref.state_dict())
    x = torch.randn(B, K, device="cuda").to(torch.bfloat16)
    gy = torch.randn(B, 1, device="cuda") / B
    outs = {}
    for name, mod in (("ref", ref), ("fast", fast)):
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = mod(xi)
        (y.float() * gy).sum().backward()
        outs[name] = (y.float(), mod.weight.grad.clone(), mod.bias.grad.clone(), xi.grad.float().clone())
    exact_y = x.double() @ ref.weight.double().t() + ref.bias.double()
    assert (outs["fast"][0].double() - exact_y).abs().max() <= (outs["ref"][0].double() - exact_y).abs().max() + 2e-2
    gw64, gb64, gx64 = gy.double().t() @ x.double(), gy.double().sum(0), gy.double() @ ref.weight.double()
    for k, exact in ((1, gw64), (2, gb64), (3, gx64)):
        scale = exact.abs().max().item()
        e_fast, e_ref = ((outs[n][k].double() - exact).abs().max().item() / scale for n in ("fast", "ref"))
        assert e_fast < 2e-2 and e_fast <= 1.5 * e_ref + 1e-3, (k, e_fast, e_ref)


def test_policy_sample_kernel_equals_model_eval_path():
    """phc_policy_sample == ModelAMPContinuous.forward(is_train=False) + value un-normalisation (amp_agent.py:309-341): same noise
    stream as torch.randn_like, same action / neglogp / sigma, un-normalised value; masked value variant."""
    from phc_amd.learning.fast_ops import policy_sample
    from phc_amd.learning.network import ModelAMPContinuous
    N, D, dev = 3001, 69, "cuda"
    g = torch.Generator(device=dev).manual_seed(2)
    mu = (torch.randn(N, D, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    value = (torch.randn(N, 1, device=dev, generator=g) * 3).to(torch.bfloat16)
    logstd = torch.full((D,), -2.9, device=dev) + torch.randn(D, device=dev, generator=g) * 0.1
    vms = RunningMeanStd((1,)).cuda().eval()
    vms.running_mean.fill_(0.7); vms.running_var.fill_(2.5)
    out = dict(a=torch.zeros(N, D, device=dev), m=torch.zeros(N, D, device=dev), s=torch.zeros(N, D, device=dev), n=torch.zeros(N, device=dev),
               v=torch.zeros(N, 1, device=dev))
    torch.manual_seed(9)
    policy_sample(mu, value, logstd, vms, out["a"], out["m"], out["s"], out["n"], out["v"])
    torch.manual_seed(9)
    muf, sigma = mu.float(), torch.exp(logstd).expand(N, D)
    action = muf + sigma * torch.randn_like(muf)
    nlp = ModelAMPContinuous.neglogp(action, muf, sigma, logstd.expand(N, D))
    torch.testing.assert_close(out["a"], action, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out["n"], nlp, rtol=1e-5, atol=1e-3)   # sum of 69 squared ~N(0,1) / sigma round-offs around 190
    assert torch.equal(out["m"], muf) and torch.allclose(out["s"], sigma)
    torch.testing.assert_close(out["v"], vms(value.float(), True), rtol=1e-6, atol=1e-6)
    mask = (torch.rand(N, device=dev) < 0.3).float()
    nv = torch.zeros(N, 1, device=dev)
    policy_sample(None, value, None, vms, None, None, None, None, nv, mask=mask)
    torch.testing.assert_close(nv, vms(value.float(), True) * (1 - mask.unsqueeze(-1)), rtol=1e-6, atol=1e-6)


def test_discriminator_loss_kernels():
    """phc_disc_bce / phc_weighted_sumsq == the torch expressions of IMAmpAgent._disc_loss (amp_agent.py:732-808)."""
    from phc_amd.learning.fast_ops import disc_bce, weighted_sumsq
    torch.manual_seed(6)
    m, dev = 1500, "cuda"
    for dtype in (torch.bfloat16, torch.float32):
        x0 = (torch.randn(3 * m, 1, device=dev) * 3).to(dtype)
        xr = x0.clone().requires_grad_(True)
        xf = xr.float()
        bce = torch.nn.BCEWithLogitsLoss()
        ref = 2.5 * 0.5 * (bce(xf[:2 * m], torch.zeros(2 * m, 1, device=dev)) + bce(xf[2 * m:], torch.ones(m, 1, device=dev)))
        ref.backward()
        xk = x0.clone().requires_grad_(True)
        loss, acc = disc_bce(xk, 2 * m, 2.5)
        loss.backward()
        torch.testing.assert_close(loss, ref.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(acc, torch.stack([(xf[:2 * m] < 0).float().mean(), (xf[2 * m:] > 0).float().mean(), xf[:2 * m].mean(), xf[2 * m:].mean()]).detach(),
                                   rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(xk.grad.float(), xr.grad.float(), rtol=1e-2 if dtype == torch.bfloat16 else 1e-5, atol=1e-9)
    ws = [torch.randn(1024, 1960, device=dev, requires_grad=True), torch.randn(512, 1024, device=dev, requires_grad=True),
          torch.randn(1, 512, device=dev, requires_grad=True)]
    coefs = [1e-4, 1e-4, 1e-4 + 0.01]
    out = weighted_sumsq(ws, coefs)
    out.backward()
    ref = sum(c * w.detach().double().square().sum() for c, w in zip(coefs, ws))
    assert abs(float(out) - float(ref)) <= 1e-5 * float(ref)
    for c, w in zip(coefs, ws):
        torch.testing.assert_close(w.grad, 2 * c * w.detach())
    g = (torch.randn(m, 1960, device=dev) * 0.1).to(torch.bfloat16).requires_grad_(True)
    pen = weighted_sumsq([g], [5.0 / m])
    pen.backward()
    ref = 5.0 * g.detach().float().square().sum(-1).mean()
    torch.testing.assert_close(pen, ref, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(g.grad.float(), (2 * 5.0 / m) * g.detach().float(), rtol=1e-2, atol=1e-9)


def test_rollout_bookkeeping_kernel_matches_torch_ops():
    """phc_rollout_bookkeeping == the torch statements of IMAmpAgent.play_steps it replaces (reference amp_agent.py:321-341)."""
    from phc_amd import _lib
    lib = _lib.load()
    torch.manual_seed(9)
    N, R = 4097, 5
    rewards = torch.randn(N, 1, device="cuda")
    dones = (torch.rand(N, device="cuda") < 0.2).long()
    term = ((torch.rand(N, device="cuda") < 0.5) & (dones > 0)).long()
    raw = torch.randn(N, R, device="cuda")
    exp_r, exp_d = torch.zeros(N, 1, device="cuda"), torch.zeros(N, device="cuda", dtype=torch.uint8)
    tf, tm = torch.rand(N, device="cuda"), torch.zeros(N, device="cuda")
    acc = torch.rand(R, device="cuda")
    cur_r, cur_l = torch.randn(N, 1, device="cuda"), torch.rand(N, device="cuda") * 50
    want_tf = tf + term.float()
    want_acc = acc + raw.mean(dim=0)
    nd = 1.0 - dones.float()
    want_r, want_l = (cur_r + rewards) * nd[:, None], (cur_l + 1) * nd
    rc = lib.phc_rollout_bookkeeping(rewards.data_ptr(), 0.5, dones.data_ptr(), term.data_ptr(), raw.data_ptr(), R, N, exp_r.data_ptr(), exp_d.data_ptr(),
                                     tf.data_ptr(), tm.data_ptr(), acc.data_ptr(), cur_r.data_ptr(), cur_l.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(exp_r, rewards * 0.5) and torch.equal(exp_d, dones.to(torch.uint8)) and torch.equal(tm, term.float()) and torch.equal(tf, want_tf)
    assert torch.equal(cur_r, want_r) and torch.equal(cur_l, want_l)
    assert torch.allclose(acc, want_acc, atol=1e-6)


@pytest.mark.parametrize("rows,cols,gated", [(257, 934, False), (64, 69, True), (1024, 512, True), (69, 512, False)])
def test_split3_kernel_writes_head_tail_chunks_of_the_operand(rows, cols, gated):
    """phc_split3_bf16 (ABI 37): head = bf16(x), tail = bf16(x - head), the three chunks in both orders and both layouts, zero padding, the ReLU gate."""
    from phc_amd.learning import fast_ops as F
    g = torch.Generator(device="cuda").manual_seed(rows * 1000 + cols)
    x = torch.randn(rows, cols + 5, device="cuda", generator=g)[:, :cols] * 3.0        # (a strided view: rows cols + 5 apart)
    gate = torch.randn(rows, cols, device="cuda", generator=g) if gated else None
    xm = x if gate is None else x * (gate > 0)
    h = xm.to(torch.bfloat16)
    l = (xm - h.float()).to(torch.bfloat16)
    assert float(((h.float() + l.float()) - xm).abs().max()) <= float(xm.abs().max()) * 2.0 ** -16
    Rp, Cp = rows + 27, F._pad32(cols)
    for order, chunks in ((0, (h, h, l)), (1, (h, l, h))):
        for chunk_major in (False, True):
            out = F._split3(x, order, gate=gate, rows_pad=Rp, cols_pad=Cp, chunk_major=chunk_major)
            if not chunk_major:
                out = out.permute(1, 0, 2)
            assert out.shape == (3, Rp, Cp)
            for c in range(3):
                assert torch.equal(out[c, :rows, :cols], chunks[c])
            assert not bool(out[:, rows:, :].any()) and not bool(out[:, :, cols:].any())
    # the ones / bias column (column `cols` of the valid rows; the forward product's bias and the weight-gradient product's bias-gradient row)
    Cq = F._pad32(cols + 1)
    e = torch.randn(rows, device="cuda", generator=g)
    eh = e.to(torch.bfloat16)
    el = (e - eh.float()).to(torch.bfloat16)
    o1 = F._split3(x, 1, gate=gate, rows_pad=Rp, cols_pad=Cq, ones=True).permute(1, 0, 2)
    o2 = F._split3(x, 0, gate=gate, rows_pad=Rp, cols_pad=Cq, extra=e).permute(1, 0, 2)
    one = torch.ones(rows, device="cuda", dtype=torch.bfloat16)
    for c, (a, b) in enumerate(zip((one, 0 * one, one), (eh, eh, el))):
        assert torch.equal(o1[c, :rows, cols], a) and torch.equal(o2[c, :rows, cols], b)
        assert torch.equal(o1[c, :rows, :cols], (h, l, h)[c]) and torch.equal(o2[c, :rows, :cols], (h, h, l)[c])
    assert not bool(o1[:, rows:, :].any()) and not bool(o1[:, :, cols + 1:].any()) and not bool(o2[:, rows:, :].any()) and not bool(o2[:, :, cols + 1:].any())


@pytest.mark.parametrize("B,K,N,relu", [(16384, 934, 1024, True), (4096, 512, 69, False), (300, 100, 40, True)])
def test_split_precision_layer_matches_the_fp64_layer(B, K, N, relu):
    """FastLinear.split_precision: output, input gradient, weight and bias gradients against the same layer in double (error 2^-16-ish, not bf16's 2^-8)."""
    from phc_amd.learning.fast_ops import FastLinear
    torch.manual_seed(B + K)
    lin = FastLinear(K, N).cuda()
    lin.split_precision, lin.fuse_relu = True, relu
    x = torch.randn(B, K, device="cuda", requires_grad=True)
    gy = torch.randn(B, N, device="cuda") / B
    y = lin(x)
    assert y.dtype == torch.float32 and y.shape == (B, N)
    y.backward(gy)
    xd = x.detach().double().requires_grad_(True)
    wd, bd = lin.weight.detach().double().requires_grad_(True), lin.bias.detach().double().requires_grad_(True)
    yd = torch.nn.functional.linear(xd, wd, bd)
    if relu:
        yd = yd * (y.detach() > 0)       # the layer's own mask: an output within 1e-5 of zero may land on the other side, and ONE flipped term moves an input gradient by percents
    yd.backward(gy.double())
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    errs = {"y": rel(y.detach(), yd.detach()), "gx": rel(x.grad, xd.grad), "gw": rel(lin.weight.grad, wd.grad), "gb": rel(lin.bias.grad, bd.grad)}
    print(f"split-precision layer {B}x{K}->{N}: max error / max value {errs}")
    assert max(errs.values()) < 5e-5, errs          # (measured 6e-6; bf16 operands: ~4e-3)
    with torch.no_grad():
        assert torch.equal(lin(x.detach()), y.detach())      # the no-grad path (rollout inference) is the same product
