import sys, os, numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
for p in ('', 'tests', 'oracle'):
    sys.path.insert(0, os.path.join(ROOT, p))
import test_learner_parity as T
from phc_amd.learning import fast_ops
import torch.nn.functional as F

def golden(name):
    return dict(np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'), allow_pickle=True))

def r(x): return x.bfloat16().float()

def run(fixture, mode):
    g = golden(fixture)
    agent = T._agent_from_golden(g)
    net = agent.model.a2c_network
    mods = [m for m in net.actor_mlp if isinstance(m, torch.nn.Linear)]
    def mk(m, head):
        def fwd(x):
            if mode == 'fp32': return F.linear(x, m.weight, m.bias)
            if not head:
                return r(F.linear(r(x), r(m.weight), r(m.bias)))
            if mode == 'bf16':       return r(F.linear(r(x), r(m.weight), r(m.bias)))
            if mode == 'bf16_f32out': return F.linear(r(x), r(m.weight), r(m.bias))
            if mode == 'head_fp32':  return F.linear(r(x), m.weight, m.bias)
            if mode == 'head_split': 
                wh = r(m.weight); wl = r(m.weight - wh)
                return F.linear(r(x), wh) + F.linear(r(x), wl) + m.bias
        return fwd
    for m in mods: m.forward = mk(m, False)
    net.mu.forward = mk(net.mu, True)
    # obs normaliser output is bf16 on the device path
    if mode != 'fp32':
        orig = agent._preproc_obs
        agent._preproc_obs = lambda *a, **k: r(orig(*a, **k))
    info = T._run_step(agent, g, 'cpu', dataset_form=False)
    params = dict(agent.model.named_parameters())
    worst = {}
    for n in params:
        if 'grad/' + n in g and ('actor_mlp' in n or 'a2c_network.mu' in n):
            ref = g['grad/' + n]; got = params[n].grad.numpy()
            worst[n] = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12))
    return max(worst.values()), float(info['actor_loss']), float(g['res/actor_loss']), float(info['kl']), float(g['res/kl'])

for fx in ('learner_step', 'learner_step_policy_actions', 'learner_step_wide'):
    for mode in ('fp32', 'bf16', 'bf16_f32out', 'head_fp32', 'head_split'):
        print(fx, mode, run(fx, mode))
