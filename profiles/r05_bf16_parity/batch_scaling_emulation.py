import os, sys, numpy as np, torch, math
import torch.nn.functional as F
torch.manual_seed(0)
g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden', 'learner_step_wide.npz'), allow_pickle=True))
W = {k[6:]: torch.from_numpy(g[k]) for k in g if k.startswith('model/a2c_network.actor_mlp') or k.startswith('model/a2c_network.mu')}
names = sorted({k.rsplit('.',1)[0] for k in W})
layers = [n for n in names if 'actor_mlp' in n] + ['a2c_network.mu']
print(layers, [tuple(W[n+'.weight'].shape) for n in layers])
r = lambda x: x.bfloat16().float()
def fwd(x, params, mode):
    h = x if mode == 'fp32' else r(x)
    for i, n in enumerate(layers):
        w, b = params[n+'.weight'], params[n+'.bias']
        last = i == len(layers) - 1
        if mode == 'fp32': h = F.linear(h, w, b)
        elif mode == 'bf16' or not last: h = r(F.linear(r(h), r(w), r(b)))
        else: h = F.linear(r(h), w, b)          # fp32 head
        if not last: h = torch.relu(h)
    return h
O = W[layers[0]+'.weight'].shape[1]; A = W['a2c_network.mu.weight'].shape[0]
sigma = math.exp(-2.9)
def grads(B, mode, data):
    params = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    x, act, old_nlp, adv = data
    mu = fwd(x, params, mode)
    nlp = 0.5 * (((act - mu) / sigma) ** 2).sum(-1)
    ratio = torch.exp(old_nlp - nlp)
    loss = torch.max(-adv * ratio, -adv * ratio.clamp(0.8, 1.2)).mean()
    loss.backward()
    return {k: v.grad for k, v in params.items()}
for B in (256, 1024, 4096, 16384):
    x = torch.randn(B, O).clamp(-5, 5)
    with torch.no_grad():
        mu0 = fwd(x, W, 'fp32')
        old_mu = mu0 + 0.3 * sigma * torch.randn_like(mu0)
        act = old_mu + sigma * torch.randn_like(mu0)
        old_nlp = 0.5 * (((act - old_mu) / sigma) ** 2).sum(-1)
        adv = torch.randn(B)
    data = (x, act, old_nlp, adv)
    ref = grads(B, 'fp32', data)
    for mode in ('bf16', 'head_fp32'):
        got = grads(B, mode, data)
        worst = max(float((got[k] - ref[k]).abs().max() / ref[k].abs().max()) for k in ref)
        l2 = math.sqrt(sum(float(((got[k] - ref[k]) ** 2).sum()) for k in ref) / sum(float((ref[k] ** 2).sum()) for k in ref))
        print(f'B={B:6d} {mode:10s} worst element / scale {worst:.4f}   relative L2 of the whole actor gradient {l2:.4f}')
