"""P4-P10: the PPO + AMP learner (`im_amp`), PyTorch-ROCm on the same device as the env.

Follows the reference's agent chain `IMAmpAgent -> AMPAgent -> CommonAgent` (phc/learning/im_amp.py,
amp_agent.py, common_agent.py; base classes from rl-games 1.1.4 are not vendored there -- their behaviour is
restated from PHC's call sites):

  train_epoch (amp_agent.py:413-504)
    play_steps (:309-397)           rollout of `horizon_length` steps, 3 MLP forwards per step
    _calc_amp_rewards (:859-878)    -log(max(1-sigmoid(D),1e-4)) * disc_reward_scale
    _combine_rewards (:848-853)     task_reward_w * r_task + disc_reward_w * r_disc
    discount_values (common_agent.py:493-505)   GAE -> `phc_gae` HIP kernel
    prepare_dataset (:399-411, common_agent.py:357-398)  advantage / value normalisation
    calc_gradients (:554-688)       PPO clip + critic + bound + discriminator (BCE, logit reg, grad penalty, weight decay)
    Adam (lr 2e-5), clip_grad_norm_(50)

MI355X-first differences (results-preserving):
  * GEMMs run in bf16 on MFMA under torch.autocast; parameters, Adam state and all losses stay fp32;
    running statistics stay fp64 (running_mean_std.py).
  * done envs are reset on the device from `reset_buf` (`task.reset_done()`), no `.nonzero()` host sync per step
    (set `faithful_reset=True` to get the reference's `env.reset(done_indices)` call sequence and RNG stream).
  * data parallelism: one process per GPU, envs sharded, ONE all-reduce (mean) of a single flat fp32 gradient
    bucket per optimizer step over RCCL (`FlatGradBucket`); it replaces horovod's `optimizer.synchronize()`
    (amp_agent.py:667-668).  Running-stat moments are averaged once per epoch (`hvd.sync_stats`).
  * on the device everything of the update that is not a GEMM runs as HIP kernels (csrc/phc_learn.hip via learning/fast_ops.py,
    DESIGN.md 4.3): observation normalisers incl. the minibatch gather, actor / critic and discriminator losses with their
    gradients, bias and split-K weight gradients of the layers, clip + Adam on the flat parameter, the rollout's sampling step;
    optionally (`hip_graph`) the forward / backward of a step is replayed from one captured hipGraph.  The torch expressions in
    this file (`_ppo_loss_torch`, `_disc_loss`, RunningMeanStd's CPU branch) remain the CPU path and the definition those kernels
    are tested against.
"""
import copy
import os
import sys
import time
import warnings

import numpy as np
import torch
from torch import nn

from .. import _lib as L
from .network import A2CMCPNetwork, A2CNetwork, A2CPNNNetwork, ModelAMPContinuous, policy_kl
from .replay_buffer import ReplayBuffer
from .fast_ops import adam_clip_step, deferred_colsums, disc_bce, input_grad_only, param_grad_only, policy_sample, ppo_loss, rows_with_grad, weighted_sumsq
from .running_mean_std import RunningMeanStd


def swap_and_flatten01(arr):
    """rl_games a2c_common.swap_and_flatten01: [T, N, ...] -> [N*T, ...] (env-major)."""
    if arr is None:
        return arr
    s = arr.size()
    return arr.transpose(0, 1).reshape(s[0] * s[1], *s[2:])


def discount_values(mb_fdones, mb_values, mb_rewards, mb_next_values, gamma, tau):
    """P5 GAE (common_agent.py:493-505).  [T,N,1] tensors.  On the HIP device this is one `phc_gae` launch;
    for CPU tensors (unit tests of the learner logic under gloo) the same recurrence runs as torch ops."""
    T, N = mb_rewards.shape[0], mb_rewards.shape[1]
    advs = torch.empty_like(mb_rewards)
    if mb_rewards.is_cuda:
        fd = mb_fdones.reshape(T, N).float().contiguous()
        v, r, nv = (x.reshape(T, N).float().contiguous() for x in (mb_values, mb_rewards, mb_next_values))
        out = advs.view(T, N)
        L.check(L.load().phc_gae(T, N, fd.data_ptr(), v.data_ptr(), r.data_ptr(), nv.data_ptr(), float(gamma), float(tau), out.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream), "phc_gae")
        return advs
    last = 0
    for t in reversed(range(T)):
        not_done = (1.0 - mb_fdones[t].float()).reshape(N, 1)
        delta = mb_rewards[t] + gamma * mb_next_values[t] - mb_values[t]
        last = delta + gamma * tau * not_done * last
        advs[t] = last
    return advs


class _marker:
    """roctx range (shows up in `rocprofv3 --marker-trace`) around the two phases of an epoch; a no-op off the device."""

    usable = True

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.on = False
        if _marker.usable and torch.cuda.is_available():
            try:
                torch.cuda.nvtx.range_push(self.name)
                self.on = True
            except Exception:   # no roctx in this build: markers off, training unaffected
                _marker.usable = False

    def __exit__(self, *a):
        if self.on:
            torch.cuda.nvtx.range_pop()


# K-padded weights are strided views with strided gradients of the SAME layout: autograd's "gradient layout contract" note does not apply
warnings.filterwarnings("ignore", message="grad and param do not obey the gradient layout contract")


VERIFIED_GRAPH_STACKS = (("2.10", "7."),)   # (torch.__version__ prefix, torch.version.hip prefix)
_graph_default_warned = [False]


def graph_default_on():
    """Is the captured (hipGraph) update the default on this software stack?  See IMAmpAgent.__init__."""
    hip = getattr(torch.version, "hip", None) or ""
    ok = any(torch.__version__.startswith(a) and hip.startswith(b) for a, b in VERIFIED_GRAPH_STACKS)
    if not ok and torch.cuda.is_available() and not _graph_default_warned[0]:
        _graph_default_warned[0] = True
        print(f"[phc_amd] torch {torch.__version__} / HIP {hip or None} is not a stack the captured update has been verified on: hipGraphs are OFF by default "
              f"(eager launches); +learning.params.config.hip_graph=True turns them on", file=sys.stderr)
    return ok


class FlatGradBucket:
    """All trainable parameters AND their gradients as views of two flat fp32 buffers.

    * one all-reduce per optimizer step (22.1 MB for `im`, 78 MB for `im_big`): over the 7 point-to-point xGMI links of
      an MI355X node RCCL moves that in well under a millisecond, so there is nothing to overlap with and one bucket is
      the right size;
    * gradient clipping is one norm + one scale of the flat gradient (`clip_grad_norm_` semantics, max-norm 50);
    * Adam updates ONE tensor (the flat parameter) instead of 22 -- three launches instead of a foreach over all of them.
    Modules keep seeing their own `weight` / `bias` (views), so state_dicts are unchanged."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        # K-padded weights (round 2): a first-layer weight [N, K] whose K is no GEMM-friendly multiple (obs 934, AMP obs 1960) is STORED with
        # rows Kp = `_pad_cols` apart (network.build_mlp tags it); the module's parameter is the strided view [:, :K] -- shapes, state dicts
        # and checkpoints are unchanged -- while the GEMMs read the padded bf16 copy against K-padded inputs (pad = 0: same numbers; the
        # first-layer forward / weight-gradient GEMMs run 20-30 % faster, scripts/probes/gemm_pad_probe.py).  The pad elements start at zero and only
        # ever see zero gradients.
        def alloc(p):
            kp = int(getattr(p, "_pad_cols", 0))
            return p.shape[0] * kp if (dev.type == "cuda" and p.dim() == 2 and kp > p.shape[1]) else p.numel()
        self.segments = []
        n = 0
        for p in self.params:
            n = (n + 3) // 4 * 4          # segments start 16-byte aligned (vector loads / stores of the gradient kernels); the gaps stay zero
            self.segments.append((n, alloc(p)))
            n += alloc(p)
        self.flat_param = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        for p, (o, k) in zip(self.params, self.segments):
            if k != p.numel():
                kp = k // p.shape[0]
                pv, gv = self.flat_param[o:o + k].view(p.shape[0], kp), self.flat[o:o + k].view(p.shape[0], kp)
                pv[:, :p.shape[1]].copy_(p.data)
                p._padded, p._grad_padded = pv, gv
                p.data, p.grad = pv[:, :p.shape[1]], gv[:, :p.shape[1]]
            else:
                self.flat_param[o:o + k].copy_(p.data.reshape(-1))
                p.data = self.flat_param[o:o + k].view_as(p)
                p.grad = self.flat[o:o + k].view_as(p)
        self.flat_param.grad = self.flat
        self.gen = 0   # bumped by zero(): lets a layer's backward store the first gradient of a step directly (fast_ops._first_write)
        for p in self.params:
            p._bucket = self
        # bf16 copy of the flat parameter for the GEMMs of the update phase: refreshed on entering shadow_scope(), then kept current
        # by the optimizer kernel; `_shadow_live` is the flag the layers look at (fast_ops.FastLinear)
        self.shadow, self._shadow_live = None, [False]
        if dev.type == "cuda":
            self.shadow = torch.zeros(n, device=dev, dtype=torch.bfloat16)
            for p, (o, k) in zip(self.params, self.segments):   # (a K-padded weight's bf16 copy is the PADDED matrix)
                sh = self.shadow[o:o + k]
                p._bf16_shadow, p._shadow_live = (sh.view(p.shape[0], k // p.shape[0]) if k != p.numel() else sh.view_as(p)), self._shadow_live

    def shadow_scope(self):
        """Context of the minibatch loop: inside it only the optimizer kernel changes the parameters, so their bf16 copies stay valid."""
        bucket = self

        class _Scope:
            def __enter__(self_):
                if bucket.shadow is not None:
                    bucket.shadow.copy_(bucket.flat_param)
                    bucket._shadow_live[0] = True

            def __exit__(self_, *a):
                bucket._shadow_live[0] = False
        return _Scope()

    def param_view(self, flat, i):
        """Parameter i's slice of a flat per-element tensor (optimizer moments) in the parameter's own shape."""
        p, (o, k) = self.params[i], self.segments[i]
        return flat[o:o + k].view(p.shape[0], k // p.shape[0])[:, :p.shape[1]] if k != p.numel() else flat[o:o + k].view_as(p)

    def spans_of(self, params):
        """([(lo, hi)] of the flat gradient that holds `params`, [(lo, hi)] of everything else): `params` must be consecutive in the bucket."""
        ids = {id(p) for p in params}
        idx = [i for i, p in enumerate(self.params) if id(p) in ids]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            raise ValueError("parameters are not one consecutive run of the gradient bucket")
        lo, hi, n = self.segments[idx[0]][0], self.segments[idx[-1]][0] + self.segments[idx[-1]][1], self.flat.numel()
        return [(lo, hi)], [(a, b) for a, b in ((0, lo), (hi, n)) if b > a]

    def zero(self, decay=None, spans=None):
        """`decay` [(parameter, c)]: start that parameter's gradient at c * parameter (an L2 term's gradient written in place instead of
        being handed to autograd as a tensor to add).  `spans`: only these ranges of the flat gradient (a pass that owns a part of the
        parameters: the other pass zeroes its own, possibly at the same time on another stream)."""
        if spans is None:
            self.flat.zero_()
        else:
            for lo, hi in spans:
                self.flat[lo:hi].zero_()
        self.gen += 1
        for p, c in decay or ():
            torch.mul(p.data, c, out=p.grad)
            p._grad_gen = self.gen

    def all_reduce_mean(self, dist, force=False, timing=None, spans=None):
        """The path's one collective.  `force`: issue it also in a one-rank group (exercises RCCL + the captured update on a 1-GPU box);
        `timing`: a list that receives (start, end) event pairs on the launch stream (bench.py: per-all-reduce time); `spans` [(lo, hi)]: only these ranges of
        the flat gradient, one all-reduce each (`split_allreduce`: the discriminator's part on its stream while the policy pass still runs).  -> number issued."""
        if dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force):
            ev = None
            if timing is not None and self.flat.is_cuda and len(timing) < 4096:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            views = [self.flat] if spans is None else [self.flat[lo:hi] for lo, hi in spans]
            for v in views:
                dist.all_reduce(v, op=dist.ReduceOp.SUM)
                if dist.get_world_size() > 1:
                    v.div_(dist.get_world_size())
            if ev is not None:
                ev[1].record()
                timing.append(ev)
            return len(views)
        return 0

    def clip_grad_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ on the flat view: same total norm, same clip coefficient (clamped to 1)."""
        total = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
        return total


class IMAmpAgent:
    def __init__(self, vec_env, cfg, dist=None, faithful_reset=False, bf16=True):
        self.vec_env = vec_env
        self.task = vec_env.task
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.multi_gpu = self.world > 1
        self.faithful_reset = faithful_reset
        params = cfg["learning"]["params"]
        c = params["config"]
        # `force_collectives`: run the gradient all-reduce also in a one-rank process group (tests: RCCL next to the captured update on one GPU)
        self._force_collectives = bool(c.get("force_collectives", False))
        self.allreduce_timing = None   # bench.py sets a list: (start, end) events of every gradient all-reduce
        self._trace = [] if c.get("trace_minibatches", False) else None
        self._step_in_epoch = 0
        # diagnostic (round 6): True = rounds 1-5's behaviour, the envs that finished on the last step of a rollout are reset at step 0 of the next one
        self._reset_at_rollout_start = bool(c.get("debug_reset_at_rollout_start", False))
        self.num_collectives = 0       # gradient all-reduces issued so far
        self.config = c
        self.device = self.task.device if hasattr(self.task, "device") else "cpu"
        self.ppo_device = self.device
        self.bf16 = bf16 and str(self.device).startswith("cuda")
        self.num_actors = vec_env.num_envs
        self.horizon_length = c["horizon_length"]
        # rl_games' A2CBase.preprocess_actions (a dependency of the reference, not vendored; common_agent.py:47 reads the switch): the env gets
        # the sampled action clamped to the action space [-1, 1]; the experience buffer keeps the unclamped sample
        self.clip_actions = bool(c.get("clip_actions", True))
        self.batch_size = self.horizon_length * self.num_actors
        self.minibatch_size = min(c["minibatch_size"], self.batch_size)
        assert self.batch_size % self.minibatch_size == 0
        self.num_minibatches = self.batch_size // self.minibatch_size
        self.mini_epochs_num = c["mini_epochs"]
        # (Rounds 5 / 6, profiles/r06_multi_clip/README.md: on the 64-clip synthetic library every run first settles on a ~13-step plateau -- lean into the first reference
        # frames, fall -- and leaves it at a SEED-dependent epoch; more optimizer steps per rollout make the escape later / rarer, they do not forbid it.  The shipped
        # 3072 envs x 36 steps learn the library for most seeds; no notice is printed any more.)
        self.gamma, self.tau = c["gamma"], c["tau"]
        self.e_clip, self.critic_coef, self.entropy_coef = c["e_clip"], c["critic_coef"], c["entropy_coef"]
        self.bounds_loss_coef = c.get("bounds_loss_coef", None)
        self.clip_value = c["clip_value"]
        self.truncate_grads, self.grad_norm = c["truncate_grads"], c["grad_norm"]
        self.normalize_input, self.normalize_value = c["normalize_input"], c["normalize_value"]
        self.normalize_advantage = c["normalize_advantage"]
        self.last_lr = float(c["learning_rate"])
        self.reward_scale = c.get("reward_shaper", {}).get("scale_value", 1)
        # AMP (_load_config_params amp_agent.py:690-707)
        self._task_reward_w, self._disc_reward_w = c["task_reward_w"], c["disc_reward_w"]
        self._amp_batch_size = int(c["amp_batch_size"])
        self._amp_minibatch_size = min(int(c["amp_minibatch_size"]), self.minibatch_size)
        self._disc_coef, self._disc_logit_reg = c["disc_coef"], c["disc_logit_reg"]
        self._disc_grad_penalty, self._disc_weight_decay = c["disc_grad_penalty"], c["disc_weight_decay"]
        self._disc_reward_scale = c["disc_reward_scale"]
        self._normalize_amp_input = c.get("normalize_amp_input", True)
        self.temp_running_mean = getattr(self.task, "temp_running_mean", True)

        obs_dim = vec_env.num_obs
        amp_dim = self.task.get_num_amp_obs()
        self.obs_shape, self.actions_num = (obs_dim,), vec_env.num_actions
        net_name = params["network"].get("name", "amp")
        if net_name == "amp":
            net = A2CNetwork(params["network"], self.actions_num, (obs_dim,), (amp_dim,))
        elif net_name == "amp_pnn":  # run_hydra.py:259: model_builder.register_network('amp_pnn', ...)
            net = A2CPNNNetwork(params["network"], self.actions_num, (obs_dim,), (amp_dim,), self.task.get_task_obs_size_detail())
        elif net_name == "amp_mcp":  # run_hydra.py:258
            net = A2CMCPNetwork(params["network"], self.actions_num, (obs_dim,), (amp_dim,), self.task.get_task_obs_size_detail())
        else:
            raise NotImplementedError(f"network '{net_name}' is not built")
        self.model = ModelAMPContinuous(net).to(self.device)
        if self.multi_gpu:  # hvd.setup_algo: broadcast rank 0's initial parameters (common_agent.py:112-113)
            for p in self.model.parameters():
                dist.broadcast(p.data, src=0)
        self.running_mean_std = RunningMeanStd((obs_dim,)).to(self.device) if self.normalize_input else None
        self.value_mean_std = RunningMeanStd((1,)).to(self.device) if self.normalize_value else None
        self._amp_input_mean_std = RunningMeanStd((amp_dim,)).to(self.device) if self._normalize_amp_input else None
        self.running_mean_std_temp = None
        # `actor_precision=split_bf16` (round 6): the actor's layers work on fp32 activations with three bf16 GEMMs per product (fast_ops._SplitLinearFn)
        # -- an option between the bf16 path (actor-gradient parity 8e-2 on the small fixtures) and fp32 GEMMs for everything (4.8 x the update)
        self._actor_split = str(c.get("actor_precision", "bf16")) == "split_bf16" and self.bf16
        if str(c.get("actor_precision", "bf16")) not in ("bf16", "split_bf16"):
            raise ValueError(f"learning.params.config.actor_precision must be bf16 or split_bf16, not {c.get('actor_precision')!r}")
        if self._actor_split:
            if net_name != "amp" or not self.normalize_input or not self.temp_running_mean:
                raise NotImplementedError("actor_precision=split_bf16 is built for the plain `amp` network with normalize_input and temp_running_mean")
            from .fast_ops import FastLinear
            for m in list(net.actor_mlp.modules()) + [net.mu]:
                if isinstance(m, FastLinear):
                    m.split_precision = True
                    m.weight._pad_cols = 0      # (the split layers read the fp32 master weights: no K-padded bf16 copy)
        self.grads = FlatGradBucket(self.model.parameters())
        # K-padded first layers: width of the padded input buffers the normalisers write (0: no padding; see FlatGradBucket)
        padded_k = {p.shape[1]: p._padded.shape[1] for p in self.grads.params if getattr(p, "_padded", None) is not None}
        self._obs_pad_cols = padded_k.get(obs_dim, 0)
        self._amp_pad_cols = padded_k.get(amp_dim, 0)
        # ON by default on the device since round 4 (`+learning.params.config.hip_graph=False` turns it off): a user of `python -m phc_amd.run` gets the
        # update the bench line quotes.  Known hazard (scripts/probes/graph_repro2.py): if the caller keeps an autograd-tracked copy of a
        # parameter alive (`w0 = p.clone()` instead of `p.detach().clone()`), that parameter's AccumulateGrad node lives on the stream it was
        # created on; the captured backward then has to hand the gradient to a stream that is not capturing and `hipStreamEndCapture` segfaults on
        # ROCm 7.2 instead of raising -- `_stale_grad_accumulators()` detects exactly that before capturing and the update falls back to eager
        # launches, as it does when a capture raises or the minibatch is small (< 2048 rows: not launch-bound).
        # the known failure mode of a bad capture on this stack is a crash, not an exception, so the DEFAULT is on only for the (torch, HIP)
        # pairs the -m gpu suite has run the captured update on (VERIFIED_GRAPH_STACKS: tests/test_env_gpu.py::test_update_graph_equals_eager_launches,
        # ::test_captured_update_trains_the_other_network_and_observation_variants incl. PNN / im_big / im_pnn_big); an explicit
        # `learning.params.config.hip_graph=True|False` always wins, `PHC_NO_GRAPH=1` / `PHC_NO_BRANCH_STREAMS=1` switch off at run time
        self._use_graph = bool(c["hip_graph"]) if "hip_graph" in c else graph_default_on()
        # The discriminator's share of an optimizer step (three input normalisers, forward, loss terms, the gradient penalty's double backward,
        # backward) shares nothing with actor + critic but the optimizer launches at the end: two chains of ~256-workgroup GEMMs and 5 us
        # finishing launches.  On the device the discriminator pass runs on its own HIP stream (_fwd_bwd); captured, it is its OWN linear
        # hipGraph replayed on that stream next to the policy graph (_graph_update).
        self._use_branches = bool(c.get("branch_streams", True)) and not os.environ.get("PHC_NO_BRANCH_STREAMS")
        # VERDICT r5 item 6b (UNMEASURED on multi-GPU hardware -- no node was ever available to this project; exercised by two-rank gloo and one-rank RCCL tests): the
        # gradient bucket cut at the policy / discriminator boundary, the discriminator's all-reduce issued on its stream as soon as its pass is done.  Off by default:
        # the shipped path is ONE all-reduce per optimizer step (north star); for `im` the discriminator holds 2.5 M of the 5.8 M gradient elements.
        self._split_allreduce = bool(c.get("split_allreduce", False))
        self._branches = None
        self._graph = self._g_data = self._g_idx = self._g_info = None
        self._graph_failed = False
        # one flat fp32 parameter; on the device clip + step are two HIP launches over it (fast_ops.adam_clip_step) and this object
        # only holds the state (checkpoint format unchanged)
        self.optimizer = torch.optim.Adam([self.grads.flat_param], self.last_lr, eps=1e-08, weight_decay=c.get("weight_decay", 0.0))

        T, N, dev = self.horizon_length, self.num_actors, self.device
        f = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self.exp = {"obses": f(T, N, obs_dim), "next_obses": f(T, N, obs_dim), "rewards": f(T, N, 1), "values": f(T, N, 1),
                    "next_values": f(T, N, 1), "neglogpacs": f(T, N), "dones": torch.zeros(T, N, device=dev, dtype=torch.uint8),
                    "actions": f(T, N, self.actions_num), "mus": f(T, N, self.actions_num), "sigmas": f(T, N, self.actions_num),
                    "amp_obs": f(T, N, amp_dim)}
        self._amp_obs_demo_buffer = ReplayBuffer(int(c["amp_obs_demo_buffer_size"]), dev)
        self._amp_replay_buffer = ReplayBuffer(int(c["amp_replay_buffer_size"]), dev)
        self._amp_replay_keep_prob = c["amp_replay_keep_prob"]
        self._idx_buf = torch.randperm(self.batch_size, device=dev)
        self.current_rewards = f(N, 1)
        self.current_lengths = f(N)
        self.epoch_num = 0
        self.frame = 0
        self.obs = None
        self.dones = torch.zeros(N, device=dev, dtype=torch.long)
        self.mean_rewards = []

    # ------------------------------------------------------------------ modes (rl_games set_eval / set_train + amp_agent.py:63-82)
    def _norms(self):
        return [m for m in (self.running_mean_std, self.value_mean_std, self._amp_input_mean_std) if m is not None]

    def set_eval(self):
        self.model.eval()
        for m in self._norms():
            m.eval()

    def set_train(self):
        self.model.train()
        for m in self._norms():
            m.train()

    def _autocast(self):
        return torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.bf16)

    # ------------------------------------------------------------------ observation pre-processing (amp_agent.py:535-552)
    def _preproc_obs(self, obs_batch, use_temp=False, row_index=None):
        if not self.normalize_input:
            return obs_batch if row_index is None else obs_batch[row_index]
        # bf16 runs: the normaliser writes the bf16 tensor the GEMMs read (the same values autocast's cast would produce)
        dt = torch.bfloat16 if self.bf16 else None
        out = None
        if obs_batch.is_cuda and self.bf16 and getattr(self, "_obs_pad_cols", 0) > obs_batch.shape[1]:
            # K-padded first layers (FlatGradBucket): the normaliser writes the left columns of a persistent [rows, Kp] buffer (pad = 0)
            rows = obs_batch.shape[0] if row_index is None else row_index.numel()
            buf = self._pad_buf("obs", rows, self._obs_pad_cols)
            out = buf[:, :obs_batch.shape[1]]
        if use_temp:  # statistics keep updating, the frozen copy provides the values (amp_agent.py:527-532)
            y = self.running_mean_std(obs_batch, norm_from=self.running_mean_std_temp, out_dtype=dt, row_index=row_index, out=out)
        else:
            y = self.running_mean_std(obs_batch, out_dtype=dt, row_index=row_index, out=out)
        return y if out is None else buf

    def _actor_obs(self, obs_batch, use_temp=False, row_index=None):
        """The ACTOR's input in split-precision mode: the normalised observation in fp32, from the same statistics the bf16 tensor of `_preproc_obs` comes from, WITHOUT
        folding the batch into them again (the bf16 call of the same step does that once)."""
        src = self.running_mean_std_temp if use_temp else self.running_mean_std
        was_training = src.training
        src.eval()
        try:
            return src(obs_batch, out_dtype=torch.float32, row_index=row_index)
        finally:
            src.train(was_training)

    def _pad_buf(self, name, rows, cols):
        """Persistent zero-initialised bf16 [rows, cols] buffers (one per use and row count: a captured graph keeps their address)."""
        bufs = self.__dict__.setdefault("_pad_bufs", {})
        key = (name, rows, cols)
        if key not in bufs:
            bufs[key] = torch.zeros((rows, cols), dtype=torch.bfloat16, device=self.device)
        return bufs[key]

    def _preproc_amp_obs(self, amp_obs, row_index=None):
        if not self._normalize_amp_input:
            return amp_obs if row_index is None else amp_obs[row_index]
        return self._amp_input_mean_std(amp_obs, out_dtype=torch.bfloat16 if self.bf16 else None, row_index=row_index)

    # ------------------------------------------------------------------ rollout (amp_agent.py:309-397)
    def preprocess_actions(self, actions, out=None):
        """What the env is stepped with (rl_games a2c_common.preprocess_actions; players: im_amp.py:68-74): clamp to [-1, 1] and rescale to the
        action space -- the identity here, VecTask's space is Box(-1, 1)."""
        if not self.clip_actions:
            return actions
        return torch.clamp(actions, -1.0, 1.0, out=out)

    def get_action_values(self, obs):
        processed = self._preproc_obs(obs)
        with torch.no_grad(), self._autocast():
            res = self.model({"is_train": False, "prev_actions": None, "obs": processed, "obs_actor": self._actor_obs(obs) if self._actor_split else None})
        if self.normalize_value:
            res["values"] = self.value_mean_std(res["values"], True)
        return res

    def _eval_critic(self, obs):
        processed = self._preproc_obs(obs)
        with torch.no_grad(), self._autocast():
            value = self.model.a2c_network.eval_critic(processed).float()
        if self.normalize_value:
            value = self.value_mean_std(value, True)
        return value

    def env_reset(self, env_ids=None):
        return self.vec_env.reset(env_ids)

    # ---- rollout segments as hipGraphs ----------------------------------------------------------------------------------------------
    # scripts/profile_rollout.py: the rollout is HOST-bound (32 steps: 16.0 ms of launch / Python time for ~9.6 ms of device work).  With
    # `hip_graph` the two policy segments of every rollout step -- (A) normalise + actor + critic + sampling into row n of the experience
    # buffer, (B) the copies of the step's outputs, the next-value critic and the episode bookkeeping -- are captured once per row n (their
    # inputs are the task's static buffers, their outputs rows of the persistent experience buffer) and replayed; the env itself (reset of
    # finished envs, stepper, post-physics) keeps its three eager launches, since its host side carries state (RNG counter, reset lists).
    def _rollout_graphs_enabled(self):
        return (self.exp["obses"].is_cuda and self._use_graph and not self._graph_failed and self.epoch_num >= 2 and not self.faithful_reset
                and hasattr(self.task, "reset_done") and not np.isfinite(self.vec_env.clip_obs) and not os.environ.get("PHC_NO_GRAPH")
                and not os.environ.get("PHC_NO_ROLLOUT_GRAPH"))

    def _replay(self, key, fn):
        g = self._roll_graphs.get(key)
        if g is None:
            in_group = self.dist is not None and self.dist.is_initialized()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            if getattr(self, "_roll_pool", None) is None:   # the rollout graphs replay one after the other and leave their results in the
                self._roll_pool = torch.cuda.graph_pool_handle()   # persistent experience buffers: their temporaries can share one pool
            with torch.cuda.graph(g, pool=self._roll_pool, capture_error_mode="thread_local" if in_group else "global"):
                fn()
            self._roll_graphs[key] = g
        g.replay()

    def play_steps(self):
        self.set_eval()
        e = self.exp
        task = self.task
        if getattr(self, "_terminated_flags", None) is None:
            self._terminated_flags = torch.zeros(self.num_actors, device=self.device)
            self._reward_raw_acc = None
            self._roll_graphs, self._roll_pool = {}, None
            self._env_actions = torch.empty_like(self.exp["actions"][0])
        terminated_flags = self._terminated_flags.zero_()
        if self._reward_raw_acc is not None:
            self._reward_raw_acc.zero_()
        # captured env launches hold the motion library, the parameter struct and the AMP reference table by value / raw pointer: when the task
        # says any of them moved (resample_motions(), evaluate() swapping libraries, flipped evaluation flags), every captured graph is dropped and
        # re-captured against the new objects (tests/test_env_gpu.py::test_rollout_graphs_follow_resample_motions_and_evaluate)
        gen = task.launch_generation() if hasattr(task, "launch_generation") else None
        if gen != getattr(self, "_roll_generation", None):
            if self._roll_graphs:
                torch.cuda.synchronize()
                self._roll_graphs.clear()
                # ... and their memory pool with them (round 5): with every graph of the pool destroyed its use count is zero, and capturing into the
                # same pool id again trips `use_count > 0` in HIPCachingAllocator unless empty_cache() could free the whole pool -- it cannot while any
                # tensor allocated during a capture is still referenced (robots: a Unitree H1 run crashed at epoch 101, its first resample_motions())
                self._roll_pool = None
            self._roll_generation = gen
        done_indices = []
        net = self.model.a2c_network
        fused = e["obses"].is_cuda
        graphed = fused and self._rollout_graphs_enabled()
        vnorm = self.value_mean_std if self.normalize_value else None

        def seg_policy(n):
            """(A) observation row, policy + value heads, sampling -> row n of the experience buffer."""
            if self.obs.data_ptr() != e["obses"][n].data_ptr():
                e["obses"][n].copy_(self.obs)
            processed = self._preproc_obs(self.obs)
            with self._autocast():
                mu, logstd = net.eval_actor(self._actor_obs(self.obs) if self._actor_split else processed)
                value = net.eval_critic(processed)
            if self._actor_split:
                value = value.float()       # (one dtype flag for both heads in phc_policy_sample)
            policy_sample(mu.contiguous(), value.contiguous(), (logstd[0] if logstd.dim() == 2 else logstd).float().contiguous(), vnorm,
                          e["actions"][n], e["mus"][n], e["sigmas"][n], e["neglogpacs"][n], e["values"][n])
            if self.clip_actions:
                self.preprocess_actions(e["actions"][n], out=self._env_actions)

        def seg_after(n, rewards, terminate, reward_raw):
            """(B) the step's outputs into row n, next-value critic (zeroed where the episode terminated), episode bookkeeping."""
            e["next_obses"][n].copy_(self.obs)
            if self._reward_raw_acc is None:
                self._reward_raw_acc = torch.zeros(reward_raw.shape[1], dtype=torch.float32, device=self.device)
            dev_ok = (rewards.is_cuda and rewards.dtype == torch.float32 and rewards.is_contiguous() and self.dones.dtype == torch.int64
                      and terminate.dtype == torch.int64 and reward_raw.dtype == torch.float32 and reward_raw.is_contiguous() and reward_raw.shape[1] <= 8)
            if dev_ok:   # rewards / dones / terminated flags / reward means / episode statistics in one launch (phc_rollout_bookkeeping)
                if getattr(self, "_terminated_mask", None) is None:
                    self._terminated_mask = torch.zeros(self.num_actors, dtype=torch.float32, device=self.device)
                terminated = self._terminated_mask
                L.check(L.load().phc_rollout_bookkeeping(
                    rewards.data_ptr(), float(self.reward_scale), self.dones.data_ptr(), terminate.data_ptr(), reward_raw.data_ptr(), reward_raw.shape[1],
                    self.num_actors, e["rewards"][n].data_ptr(), e["dones"][n].data_ptr(), terminated_flags.data_ptr(), terminated.data_ptr(),
                    self._reward_raw_acc.data_ptr(), self.current_rewards.data_ptr(), self.current_lengths.data_ptr(),
                    torch.cuda.current_stream().cuda_stream), "phc_rollout_bookkeeping")
            else:
                e["rewards"][n].copy_(rewards if self.reward_scale == 1 else rewards * self.reward_scale)
                e["dones"][n].copy_(self.dones)
                terminated = terminate.float()
                terminated_flags.add_(terminated)
                self._reward_raw_acc.add_(reward_raw.mean(dim=0))
            with self._autocast():
                value = net.eval_critic(self._preproc_obs(self.obs))
            policy_sample(None, value.contiguous(), None, vnorm, None, None, None, None, e["next_values"][n], mask=terminated)
            if not dev_ok:
                not_dones = 1.0 - self.dones.float()
                self.current_rewards.add_(rewards).mul_(not_dones.unsqueeze(1))
                self.current_lengths.add_(1).mul_(not_dones)

        # ONE hipGraph per rollout step (round 3): reset of the finished envs, observation row, actor / critic / sampling, stepper, post-physics, AMP
        # window copy, next-value critic and bookkeeping -- ~40 launches -- replayed with a single host call; the task's host-side state (AMP
        # window position, reset-list slot, info dict) is advanced by task.replay_step_host().  Keyed by the step index and that state; the
        # start-time draws of the captured reset launch stay fresh through the device-side call counter (phc_im_buffers_t.reset_rng_counter).
        whole = (graphed and self._reward_raw_acc is not None and getattr(task, "whole_step_capturable", lambda: False)()
                 and not os.environ.get("PHC_NO_STEP_GRAPH"))
        if whole:
            task.align_amp_window()

            def whole_step(n):
                if n > 0 or self._reset_at_rollout_start:   # (step 0 resets nothing: see the eager loop below)
                    task.reset_done()
                self.obs = task.obs_buf
                seg_policy(n)
                self.obs, rewards, self.dones, infos = self.vec_env.step(self._env_actions if self.clip_actions else e["actions"][n])
                rewards = rewards.unsqueeze(1) if rewards.dim() == 1 else rewards
                e["amp_obs"][n].copy_(infos["amp_obs"])
                seg_after(n, rewards, infos["terminate"], infos["reward_raw"])

            for n in range(self.horizon_length):
                key = ("step", n) + task.rollout_step_key()
                if key in self._roll_graphs:
                    self._roll_graphs[key].replay()
                    task.replay_step_host(reset=n > 0 or self._reset_at_rollout_start)
                else:
                    self._replay(key, lambda: whole_step(n))   # capture (runs the task's own host bookkeeping), then the first replay: no bookkeeping
            self.obs, self.dones = task.obs_buf, task.reset_buf
        for n in range(self.horizon_length if not whole else 0):
            if self.faithful_reset or not hasattr(task, "reset_done"):
                self.obs = self.env_reset(done_indices)
            else:
                # The reference starts every rollout with `done_indices = []` (amp_agent.py:313): the envs that finished on the LAST step of the previous
                # rollout are not reset at step 0 -- they run one more step, are flagged again by that step's `_compute_reset` (their clip is still over /
                # they are still down) and are reset at step 1.  Kept: one in `horizon_length` episode ends carries that extra transition in the reference's
                # data too (tests/test_learner_epoch.py pins it).
                if n > 0 or self._reset_at_rollout_start:
                    task.reset_done()
                if np.isfinite(self.vec_env.clip_obs):   # clamp straight into the experience buffer row
                    self.obs = torch.clamp(task.obs_buf, -self.vec_env.clip_obs, self.vec_env.clip_obs, out=e["obses"][n])
                else:
                    self.obs = task.obs_buf
            if fused:
                if graphed:
                    self._replay(("policy", n, self.obs.data_ptr()), lambda: seg_policy(n))
                else:
                    seg_policy(n)
                res = {"actions": self._env_actions if self.clip_actions else e["actions"][n]}
            else:
                if self.obs.data_ptr() != e["obses"][n].data_ptr():
                    e["obses"][n].copy_(self.obs)
                res = self.get_action_values(self.obs)
                for k in ("values", "neglogpacs", "actions", "mus", "sigmas"):
                    e[k][n].copy_(res[k])
                res = {"actions": self.preprocess_actions(res["actions"])}
            self.obs, rewards, self.dones, infos = self.vec_env.step(res["actions"])
            rewards = rewards.unsqueeze(1) if rewards.dim() == 1 else rewards
            e["amp_obs"][n].copy_(infos["amp_obs"])   # (eager: the window of the task's AMP strip moves every step)
            if fused:
                if self._reward_raw_acc is None:       # first rollout: creates the accumulator eagerly
                    seg_after(n, rewards, infos["terminate"], infos["reward_raw"])
                elif graphed:
                    key = ("after", n, self.obs.data_ptr(), rewards.data_ptr(), self.dones.data_ptr(), infos["terminate"].data_ptr(),
                           infos["reward_raw"].data_ptr())
                    self._replay(key, lambda: seg_after(n, rewards, infos["terminate"], infos["reward_raw"]))
                else:
                    seg_after(n, rewards, infos["terminate"], infos["reward_raw"])
            else:
                e["rewards"][n].copy_(rewards if self.reward_scale == 1 else rewards * self.reward_scale)
                e["next_obses"][n].copy_(self.obs)
                e["dones"][n].copy_(self.dones)
                terminated = infos["terminate"].float()
                terminated_flags += terminated
                rr = infos["reward_raw"].mean(dim=0)
                self._reward_raw_acc = rr.clone() if self._reward_raw_acc is None else self._reward_raw_acc.add_(rr)
                next_vals = self._eval_critic(self.obs)
                next_vals = next_vals * (1.0 - terminated.unsqueeze(-1))
                e["next_values"][n].copy_(next_vals)
                self.current_rewards += rewards
                self.current_lengths += 1
                not_dones = 1.0 - self.dones.float()
                self.current_rewards = self.current_rewards * not_dones.unsqueeze(1)
                self.current_lengths = self.current_lengths * not_dones
            if self.faithful_reset or not hasattr(task, "reset_done"):   # (a task without the device-side reset gets the reference's index list)
                done_indices = self.dones.nonzero(as_tuple=False)[:, 0]
        reward_raw = self._reward_raw_acc
        mb_fdones = e["dones"].float()
        amp_rewards = self._calc_amp_rewards(e["amp_obs"])
        mb_rewards = self._combine_rewards(e["rewards"], amp_rewards)
        mb_advs = discount_values(mb_fdones, e["values"], mb_rewards, e["next_values"], self.gamma, self.tau)
        mb_returns = mb_advs + e["values"]
        batch = {k: swap_and_flatten01(e[k]) for k in ("obses", "rewards", "values", "neglogpacs", "dones", "actions", "mus", "sigmas", "amp_obs")}
        batch["returns"] = swap_and_flatten01(mb_returns)
        batch["terminated_flags"] = terminated_flags
        batch["reward_raw"] = reward_raw / self.horizon_length
        batch["played_frames"] = self.batch_size
        batch["disc_rewards"] = swap_and_flatten01(amp_rewards["disc_rewards"])
        batch["mb_rewards"] = swap_and_flatten01(mb_rewards)
        return batch

    def _eval_disc(self, amp_obs):
        with self._autocast():
            return self.model.a2c_network.eval_disc(self._preproc_amp_obs(amp_obs)).float()

    def _calc_amp_rewards(self, amp_obs):
        with torch.no_grad():
            T, N, A = amp_obs.shape
            logits = self._eval_disc(amp_obs.reshape(T * N, A)).reshape(T, N, 1)
            prob = 1 / (1 + torch.exp(-logits))
            disc_r = -torch.log(torch.maximum(1 - prob, torch.tensor(0.0001, device=amp_obs.device)))
            disc_r = disc_r * self._disc_reward_scale
        return {"disc_rewards": disc_r}

    def _combine_rewards(self, task_rewards, amp_rewards):
        return self._task_reward_w * task_rewards + self._disc_reward_w * amp_rewards["disc_rewards"]

    # ------------------------------------------------------------------ dataset (common_agent.py:357-398,589-599; amp_agent.py:399-411)
    def prepare_dataset(self, b):
        advantages = torch.sum(b["returns"] - b["values"], axis=1)
        if self.normalize_advantage:
            advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)  # per rank, as the reference (common_agent.py:596-597)
        values, returns = b["values"], b["returns"]
        if self.normalize_value:
            values = self.value_mean_std(values)
            returns = self.value_mean_std(returns)
        self.dataset = {"old_values": values, "old_logp_actions": b["neglogpacs"], "advantages": advantages, "returns": returns,
                        "actions": b["actions"], "obs": b["obses"], "mu": b["mus"], "sigma": b["sigmas"], "amp_obs": b["amp_obs"],
                        "amp_obs_demo": b["amp_obs_demo"], "amp_obs_replay": b["amp_obs_replay"]}

    def _get_item(self, idx):
        s, e = idx * self.minibatch_size, (idx + 1) * self.minibatch_size
        sample_idx = self._idx_buf[s:e]
        amp_idx = sample_idx[:self._amp_minibatch_size]  # calc_gradients only reads rows [0:amp_minibatch_size] (amp_agent.py:570-577)
        if sample_idx.is_cuda:
            # the device kernels read the dataset rows in place (phc_running_norm / phc_ppo_loss take the row index): nothing is gathered
            sample_idx = sample_idx.clone()   # _idx_buf is re-drawn below, before this minibatch's kernels have run
            out = {"_dataset": self.dataset, "_idx": sample_idx, "_amp_idx": sample_idx[:self._amp_minibatch_size]}
        else:
            out = {k: v[amp_idx if k.startswith("amp_obs") else sample_idx] for k, v in self.dataset.items()}
        if e >= self.batch_size:
            self._idx_buf[:] = torch.randperm(self.batch_size, device=self._idx_buf.device)
        return out

    # ------------------------------------------------------------------ losses (common_agent.py:512-520,564-587; amp_agent.py:732-808)
    def bound_loss(self, mu):
        if self.bounds_loss_coef is None:
            return torch.zeros((), device=mu.device)
        soft_bound = 1.0
        hi = torch.clamp_min(mu - soft_bound, 0.0) ** 2
        lo = torch.clamp_max(mu + soft_bound, 0.0) ** 2
        return (lo + hi).sum(axis=-1)

    def _disc_loss(self, disc_agent_logit, disc_demo_logit, obs_demo):
        bce = torch.nn.BCEWithLogitsLoss()
        disc_loss = 0.5 * (bce(disc_agent_logit, torch.zeros_like(disc_agent_logit)) + bce(disc_demo_logit, torch.ones_like(disc_demo_logit)))
        net = self.model.a2c_network
        disc_logit_loss = torch.sum(torch.square(net.get_disc_logit_weights()))
        disc_loss = disc_loss + self._disc_logit_reg * disc_logit_loss
        with input_grad_only():
            grad = torch.autograd.grad(disc_demo_logit, obs_demo, grad_outputs=torch.ones_like(disc_demo_logit), create_graph=True,
                                       retain_graph=True, only_inputs=True)[0].float()   # bf16 leaf (device normaliser output): penalty in fp32
        disc_grad_penalty = torch.mean(torch.sum(torch.square(grad), dim=-1))
        disc_loss = disc_loss + self._disc_grad_penalty * disc_grad_penalty
        if self._disc_weight_decay != 0:
            w = torch.cat(net.get_disc_weights(), dim=-1)
            disc_loss = disc_loss + self._disc_weight_decay * torch.sum(torch.square(w))
        return {"disc_loss": disc_loss, "disc_grad_penalty": disc_grad_penalty.detach(), "disc_logit_loss": disc_logit_loss.detach(),
                "disc_agent_acc": (disc_agent_logit < 0).float().mean().detach(), "disc_demo_acc": (disc_demo_logit > 0).float().mean().detach(),
                # (the reference keeps the logit tensors and logs their mean over the epoch, amp_agent.py:786-787,911-912: equal-sized minibatches, same number)
                "disc_agent_logit": disc_agent_logit.detach().float().mean(), "disc_demo_logit": disc_demo_logit.detach().float().mean()}

    # Every scalar the fused loss kernels produce lands in ONE fp32 vector (`_raw`), which a graphed step adds to its accumulator
    # in one launch; the info dict is derived from it outside the step (all entries are linear in it).  Layout, nw = number of
    # discriminator weight matrices:  [0:7] phc_ppo_loss (loss, a_loss, c_loss, b_loss, entropy, kl, clip fraction) | [7:12] phc_disc_bce (k * bce,
    # agent acc, demo acc, mean agent logit, mean demo logit) | [12:13+nw] weight terms (k * (decay + logit reg), |W_i|^2 ...) |
    # [13+nw:15+nw] penalty (k * pen, |grad|^2).
    _RAW_FIXED = 15

    def _raw_buffer(self):
        nw = len(self.model.a2c_network.get_disc_weights_raw())
        if getattr(self, "_raw", None) is None or self._raw.numel() != self._RAW_FIXED + nw:
            self._raw = torch.zeros(self._RAW_FIXED + nw, dtype=torch.float32, device=self.device)
        return self._raw, nw

    def _info_from_raw(self, raw):
        nw = raw.numel() - self._RAW_FIXED
        k = self._disc_coef
        return {"actor_loss": raw[1], "critic_loss": raw[2], "b_loss": raw[3], "entropy": raw[4], "kl": raw[5], "actor_clip_frac": raw[6],
                "disc_loss": (raw[7] + raw[12] + raw[13 + nw]) / k, "disc_grad_penalty": raw[13 + nw] / (self._disc_grad_penalty * k),
                "disc_logit_loss": raw[12 + nw], "disc_agent_acc": raw[8], "disc_demo_acc": raw[9], "disc_agent_logit": raw[10], "disc_demo_logit": raw[11]}

    def _disc_loss_fused(self, logits, m, obs_demo):
        """`_disc_loss` on the device with the pieces as kernels (fast_ops.disc_bce / weighted_sumsq): every term already carries
        `disc_coef` and enters the total loss with weight one.  `logits` [3m, 1]: agent, replay, demo rows.
        -> (the three loss terms, [(weight, c)] whose gradient c * weight is to be preloaded by FlatGradBucket.zero)."""
        k = self._disc_coef
        raw, nw = self._raw_buffer()
        bce, _ = disc_bce(logits, 2 * m, k, out=raw[7:12])
        ws, coefs, preload = self._disc_decay_terms()
        # (a K-padded weight enters with its padded storage -- the pad is zero --, when its gradient is preloaded anyway)
        l2 = weighted_sumsq([getattr(w, "_padded", w) if preload else w.contiguous() for w in ws], coefs, out=raw[12:13 + nw], preloaded=preload)
        # d(sum of the demo logits) / d(demo rows): cotangent = [0; 0; 1] over the [agent; replay; demo] logits, and the layers are told
        # that only the last row block carries anything (GEMMs over m instead of 3m rows, here and in the second-order pass)
        with input_grad_only(row_start=2 * m):
            grad = torch.autograd.grad(logits, obs_demo, grad_outputs=self._demo_row_mask(m, logits), create_graph=True, retain_graph=True,
                                       only_inputs=True)[0]
        pen = weighted_sumsq([grad], [self._disc_grad_penalty * k / m], out=raw[13 + nw:15 + nw])
        return [bce, l2, pen], ([(w, 2.0 * c) for w, c in zip(ws, coefs)] if preload else None)

    def _disc_decay_terms(self):
        """(discriminator weights, coefficient of each one's sum of squares in the total loss, whether their gradients 2 c w can be preloaded
        into the flat bucket)."""
        k = self._disc_coef
        ws = [p for p in self.model.a2c_network.get_disc_weights_raw()]
        coefs = [self._disc_weight_decay * k] * len(ws)
        coefs[-1] += self._disc_logit_reg * k   # the logit layer: regulariser + weight decay
        return ws, coefs, all(getattr(w, "_bucket", None) is self.grads for w in ws)

    def _demo_row_mask(self, m, like):
        key = (m, like.dtype, like.device)
        if getattr(self, "_ones_key", None) != key:
            mask = torch.zeros((3 * m, 1), dtype=like.dtype, device=like.device)
            mask[2 * m:] = 1
            self._ones_key, self._ones = key, mask
        return self._ones

    def _fwd_bwd(self, d, want_info=True):
        """Forward + losses + backward into the flat gradient bucket (amp_agent.py:554-655); no host sync.  `want_info=False` (the
        captured step): when all scalars are in `self._raw` the info dict is not built (it would cost launches)."""
        idx = amp_idx = None
        if "_dataset" in d:   # device: minibatch = (dataset, row index); the kernels below read the rows in place
            d, idx, amp_idx = d["_dataset"], d["_idx"], d["_amp_idx"]
        br = self._branch_streams(d["obs"].device) if (d["obs"].is_cuda and self._disc_grad_penalty > 0 and self._normalize_amp_input) else None
        if br:
            # the discriminator pass on its stream next to the policy pass: see __init__
            main = torch.cuda.current_stream(d["obs"].device)
            br.wait_stream(main)
            split = self._split_active() and want_info      # (want_info=False: a captured step's warm-up pass -- gradients only, no collective)
            with torch.cuda.stream(br):
                self._disc_pass(d, amp_idx)
                if split:
                    self._grad_all_reduce("disc")
            self._policy_pass(d, idx)
            if split:
                self._grad_all_reduce("policy")
            self._reduced_in_pass = split
            main.wait_stream(br)
            return self._info_from_raw(self._raw.clone()) if want_info else None   # (a copy: the next step overwrites `_raw`)
        obs = self._preproc_obs(d["obs"], use_temp=self.temp_running_mean, row_index=idx)
        fused = obs.is_cuda
        fused_disc = fused and self._disc_grad_penalty > 0 and self._normalize_amp_input
        if fused_disc:
            # the normaliser writes the three AMP batches into the row blocks of ONE [agent; replay; demo] buffer (no torch.cat in front
            # of the discriminator); the demo block is also the leaf the gradient penalty differentiates to
            m = amp_idx.numel() if amp_idx is not None else d["amp_obs"].shape[0]
            dt = torch.bfloat16 if self.bf16 else torch.float32
            A = d["amp_obs"].shape[1]
            if self.bf16 and getattr(self, "_amp_pad_cols", 0) > A:   # K-padded discriminator input (pad columns stay zero)
                cat = self._pad_buf("amp_cat", 3 * m, self._amp_pad_cols)
            else:
                cat = torch.empty((3 * m, A), dtype=dt, device=obs.device)
            for k, key in enumerate(("amp_obs", "amp_obs_replay", "amp_obs_demo")):
                self._amp_input_mean_std(d[key], out_dtype=dt, row_index=amp_idx, out=cat[k * m:(k + 1) * m, :A])
            amp_obs = cat[:m]
            amp_obs_demo = cat[2 * m:].requires_grad_(True)
            inp = {"is_train": True, "obs": obs, "amp_obs_cat": rows_with_grad(cat, amp_obs_demo, 2 * m), "raw_disc_logits": True}
        else:
            amp_obs = self._preproc_amp_obs(d["amp_obs"], amp_idx)
            amp_obs_replay = self._preproc_amp_obs(d["amp_obs_replay"], amp_idx)
            amp_obs_demo = self._preproc_amp_obs(d["amp_obs_demo"], amp_idx)
            amp_obs_demo.requires_grad_(True)
            inp = {"is_train": True, "prev_actions": d["actions"], "obs": obs, "amp_obs": amp_obs, "amp_obs_replay": amp_obs_replay,
                   "amp_obs_demo": amp_obs_demo, "raw_disc_logits": False}
        if self._actor_split and fused:
            inp["obs_actor"] = self._actor_obs(d["obs"], use_temp=self.temp_running_mean, row_index=idx)
        with self._autocast():
            res = self.model.forward_heads(inp) if fused else self.model(inp)
        if self._actor_split and fused:
            res["value"] = res["value"].float()
        decay = None
        if fused_disc:
            roots, decay = self._disc_loss_fused(res["disc_logits"], amp_obs.shape[0], amp_obs_demo)
            disc_info = None
        else:
            disc_info = self._disc_loss(torch.cat([res["disc_agent_logit"], res["disc_agent_replay_logit"]], dim=0), res["disc_demo_logit"], amp_obs_demo)
            roots = [self._disc_coef * disc_info["disc_loss"]]
        if fused:
            # actor / critic losses and their gradients w.r.t. the two heads: one HIP pass (phc_ppo_loss) instead of ~100 launches
            ppo, st = ppo_loss(res["mu"].contiguous(), res["value"].contiguous(), res["logstd"], d["actions"], d["old_logp_actions"], d["advantages"],
                               d["returns"], d["old_values"], d["mu"], d["sigma"], self.e_clip, self.critic_coef, self.entropy_coef,
                               self.bounds_loss_coef, self.clip_value, unit_grad=True, row_index=idx, out=self._raw_buffer()[0][0:7])
            # `ppo` and the fused discriminator terms enter the total with weight one (unit_grad): they are the roots of ONE backward
            # pass with constant unit cotangents -- no sum node, no fill launches
            roots = [ppo] + roots
            info = None if fused_disc else {"actor_loss": st[0], "critic_loss": st[1], "b_loss": st[2], "entropy": st[3], "kl": st[4], "actor_clip_frac": st[5]}
            self.grads.zero(decay)
            with param_grad_only(), deferred_colsums():
                torch.autograd.backward(roots, grad_tensors=[self._unit_cotangent(r) for r in roots])
        else:
            assert idx is None
            loss, info = self._ppo_loss_torch(res, d, disc_info)
            self.grads.zero()
            loss.backward()
        if info is None:   # everything is in the raw vector
            return self._info_from_raw(self._raw.clone()) if want_info else None   # (a copy: the next step overwrites `_raw`)
        info.update({k: (v.detach() if torch.is_tensor(v) else v) for k, v in disc_info.items()})
        return info

    def _branch_streams(self, device):
        """The discriminator's stream of the optimizer step, or None (CPU, switched off)."""
        if not self._use_branches or torch.device(device).type != "cuda":
            return None
        if self._branches is None:
            from .fast_ops import register_lane
            net = self.model.a2c_network
            disc_params = [p for mod in (net._disc_mlp, net._disc_logits) for p in mod.parameters()]
            try:
                self._disc_spans, self._policy_spans = self.grads.spans_of([p for p in disc_params if getattr(p, "_bucket", None) is self.grads])
            except ValueError:      # (a network whose discriminator parameters are not one run of the bucket: one stream, as before round 4)
                self._use_branches = False
                return None
            self._branches = torch.cuda.Stream(device)
            register_lane(self._branches, "disc")      # (its kernels get their own reduction scratch buffers)
        return self._branches

    def _disc_pass(self, d, amp_idx):
        """The discriminator's share of an optimizer step on the CURRENT stream: its slice of the gradient bucket zeroed (weight-decay gradients
        preloaded), the three AMP batches normalised into one [agent; replay; demo] buffer, forward, loss terms, backward."""
        ws, coefs, preload = self._disc_decay_terms()
        self.grads.zero([(w, 2.0 * c) for w, c in zip(ws, coefs)] if preload else None, spans=self._disc_spans)
        m = amp_idx.numel() if amp_idx is not None else d["amp_obs"].shape[0]
        dt = torch.bfloat16 if self.bf16 else torch.float32
        A = d["amp_obs"].shape[1]
        if self.bf16 and getattr(self, "_amp_pad_cols", 0) > A:   # K-padded discriminator input (pad columns stay zero)
            cat = self._pad_buf("amp_cat", 3 * m, self._amp_pad_cols)
        else:
            cat = torch.empty((3 * m, A), dtype=dt, device=d["amp_obs"].device)
        for k, key in enumerate(("amp_obs", "amp_obs_replay", "amp_obs_demo")):
            self._amp_input_mean_std(d[key], out_dtype=dt, row_index=amp_idx, out=cat[k * m:(k + 1) * m, :A])
        amp_obs_demo = cat[2 * m:].requires_grad_(True)
        with self._autocast():
            logits = self.model.a2c_network.eval_disc(rows_with_grad(cat, amp_obs_demo, 2 * m))
        roots, _ = self._disc_loss_fused(logits, m, amp_obs_demo)
        with param_grad_only(), deferred_colsums():   # (bias gradients: first stages in the layers, ONE finishing launch here)
            torch.autograd.backward(roots, grad_tensors=[self._unit_cotangent(r) for r in roots])

    def _policy_pass(self, d, idx):
        """Actor + critic share of an optimizer step on the current stream: their slices of the gradient bucket zeroed, the observation
        normalised, both forward passes, the fused loss kernel, backward.  (The critic on a third stream was measured too: a graph with a
        fork inside leaves the runtime's packet-replay path -- profiles/r04_ppo/README.md.)"""
        net = self.model.a2c_network
        self.grads.zero(spans=self._policy_spans)
        obs = self._preproc_obs(d["obs"], use_temp=self.temp_running_mean, row_index=idx)
        with self._autocast():
            value = net.eval_critic(obs)
            mu, logstd = net.eval_actor(self._actor_obs(d["obs"], use_temp=self.temp_running_mean, row_index=idx) if self._actor_split else obs)
        if self._actor_split:
            value = value.float()           # (phc_ppo_loss takes both heads in one type)
        ppo, _ = ppo_loss(mu.contiguous(), value.contiguous(), logstd[0] if logstd.dim() == 2 else logstd, d["actions"], d["old_logp_actions"],
                          d["advantages"], d["returns"], d["old_values"], d["mu"], d["sigma"], self.e_clip, self.critic_coef, self.entropy_coef,
                          self.bounds_loss_coef, self.clip_value, unit_grad=True, row_index=idx, out=self._raw_buffer()[0][0:7])
        with param_grad_only(), deferred_colsums():
            torch.autograd.backward([ppo], grad_tensors=[self._unit_cotangent(ppo)])

    def _unit_cotangent(self, like):
        key = (like.dtype, like.device)
        c = getattr(self, "_unit_cot", None)
        if c is None:
            c = self._unit_cot = {}
        if key not in c:
            c[key] = torch.ones((), dtype=like.dtype, device=like.device)
        return c[key]

    def _ppo_loss_torch(self, res, d, disc_info):
        """The actor / critic losses as torch expressions (amp_agent.py:598-640): the CPU path and the definition `phc_ppo_loss` is
        tested against."""
        ratio = torch.exp(d["old_logp_actions"] - res["prev_neglogp"])
        adv = d["advantages"]
        a_loss = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - self.e_clip, 1.0 + self.e_clip))
        values, ret, vp = res["values"], d["returns"], d["old_values"]
        if self.clip_value:
            vpc = vp + (values - vp).clamp(-self.e_clip, self.e_clip)
            c_loss = torch.max((values - ret) ** 2, (vpc - ret) ** 2)
        else:
            c_loss = (ret - values) ** 2
        b_loss = self.bound_loss(res["mus"])
        a_loss, c_loss, b_loss, entropy = a_loss.mean(), c_loss.mean(), b_loss.mean(), res["entropy"].mean()
        bl = self.bounds_loss_coef if self.bounds_loss_coef is not None else 0.0
        loss = a_loss + self.critic_coef * c_loss - self.entropy_coef * entropy + bl * b_loss + self._disc_coef * disc_info["disc_loss"]
        with torch.no_grad():
            kl = policy_kl(res["mus"].detach(), res["sigmas"].detach(), d["mu"], d["sigma"])
            clip_frac = (torch.abs(ratio - 1.0) > self.e_clip).float().mean()     # common_agent.py:570-571,327
        return loss, {"actor_loss": a_loss.detach(), "critic_loss": c_loss.detach(), "b_loss": b_loss.detach(), "entropy": entropy.detach(), "kl": kl,
                      "actor_clip_frac": clip_frac}

    def _clip_and_step(self, step_device=None):
        """`step_device`: inside a captured graph -- the kernels count the optimizer step on the device, the host-side `step` is advanced by
        the caller (one per replay)."""
        if self.grads.flat.is_cuda:
            adam_clip_step(self.optimizer, self.grads.flat_param, self.grads.flat, self.grad_norm if self.truncate_grads else None,
                           shadow=self.grads.shadow, step_device=step_device, count_host=step_device is None)
            return
        if self.truncate_grads:
            self.grads.clip_grad_norm_(self.grad_norm)
        self.optimizer.step()

    def _amp_rows(self, d):
        m = self._amp_minibatch_size
        if "_dataset" in d:
            return d
        return {k: (v[0:m] if k.startswith("amp_obs") else v) for k, v in d.items()}

    def calc_gradients(self, d):
        """One optimizer step (amp_agent.py:554-688): forward/backward, the path's one collective -- the flat gradient
        all-reduce, replacing `optimizer.synchronize()` (:667-668) -- then clip + Adam on the flat parameter."""
        self.set_train()
        d = self._amp_rows(d)
        self._reduced_in_pass = False
        info = self._fwd_bwd(d)
        if not self._reduced_in_pass:
            self._grad_all_reduce()
        restore = self._debug_freeze(self._step_in_epoch) if self._debug_groups() else None
        self._clip_and_step()
        if restore is not None:
            restore()
        self._step_in_epoch += 1
        return info

    # ---- diagnostic knobs: `+learning.params.config.debug_actor_steps=K` (`debug_critic_steps`, `debug_disc_steps`) lets that network take only the
    # FIRST K optimizer steps of every epoch -- what `mini_epochs` does for all three at once, separately.  On the other steps the group's gradient is zeroed before
    # the clip and its parameters + Adam moments are put back after the step (the optimizer kernel works on the whole flat parameter).  Eager launches only.
    def _debug_groups(self):
        if getattr(self, "_dbg", None) is None:
            c, self._dbg = self.config, {}
            names = {id(p): n for n, p in self.model.named_parameters()}
            for grp, key in (("actor", "debug_actor_steps"), ("critic", "debug_critic_steps"), ("disc", "debug_disc_steps")):
                if c.get(key) is None:
                    continue
                of = lambda n: "disc" if "_disc" in n else ("critic" if ("critic" in n or ".value." in n) else "actor")
                spans = [(o, o + k) for p, (o, k) in zip(self.grads.params, self.grads.segments) if of(names[id(p)]) == grp]
                self._dbg[grp] = (int(c[key]), spans)
        return self._dbg

    def _debug_freeze(self, step_in_epoch):
        """-> restore() for the groups that sit this step out (None when every group steps)."""
        off = [spans for k, spans in self._debug_groups().values() if step_in_epoch >= k]
        if not off:
            return None
        from .fast_ops import adam_state
        st = adam_state(self.optimizer, self.grads.flat_param) if self.grads.flat.is_cuda else self.optimizer.state.get(self.grads.flat_param, {})
        bufs = [self.grads.flat_param] + [st[k] for k in ("exp_avg", "exp_avg_sq") if k in st]
        saved = []
        for spans in off:
            for lo, hi in spans:
                self.grads.flat[lo:hi].zero_()
                saved.append((lo, hi, [b[lo:hi].clone() for b in bufs]))

        def restore():
            for lo, hi, vals in saved:
                for b, v in zip(bufs, vals):
                    b[lo:hi].copy_(v)
            if self.grads.shadow is not None:
                self.grads.shadow.copy_(self.grads.flat_param)
        return restore

    def _grad_all_reduce(self, part=None):
        """`part` None: the whole flat gradient (one collective per optimizer step, the default).  "disc" / "policy" (`split_allreduce`): that pass's range of the
        bucket, issued on the CURRENT stream right behind the pass -- the discriminator's all-reduce then runs while the (longer) policy chain is still computing."""
        if self.multi_gpu or (self._force_collectives and self.dist is not None):
            spans = None if part is None else (self._disc_spans if part == "disc" else self._policy_spans)
            self.num_collectives += self.grads.all_reduce_mean(self.dist, force=self._force_collectives, timing=self.allreduce_timing, spans=spans)

    def _split_active(self):
        """Two all-reduces per optimizer step instead of one (`+learning.params.config.split_allreduce=True`; needs the two-stream step)."""
        return (self._split_allreduce and (self.multi_gpu or (self._force_collectives and self.dist is not None)) and self._branches is not None)

    # ------------------------------------------------------------------ the optimizer step as a hipGraph
    # With the loss, normaliser and optimizer kernels fused, a step is ~140 launches of 2 ms total device time and the host needs
    # 2.7-3.9 ms to issue them (it varies with the box): launch-bound.  Device runs can therefore capture the forward / backward part of
    # ONE step -- minibatch given by a row-index buffer into persistent dataset tensors -- and replay it 48 times per epoch; the
    # gradient all-reduce (multi-GPU) and the two optimizer launches follow each replay eagerly, so the graph holds no collective.
    def _graph_enabled(self):
        return (self.grads.flat.is_cuda and self._use_graph and not self._graph_failed and not self._debug_groups() and self.minibatch_size >= int(self.config.get("hip_graph_min_rows", 2048))
                and not os.environ.get("PHC_NO_GRAPH"))

    def _stale_grad_accumulators(self):
        """True if some parameter's AccumulateGrad node is kept alive by a graph outside this update (e.g. `w0 = p.clone()` held by
        the caller): such a node is bound to the stream it was created on and breaks stream capture.  A parameter caches its
        accumulator weakly, so a node nobody else holds is gone once we drop it -- a tag we leave on it tells the two cases apart."""
        from torch.autograd.graph import get_gradient_edge
        for p in self.grads.params:
            get_gradient_edge(p).node.metadata["phc_probe"] = True
        return any("phc_probe" in get_gradient_edge(p).node.metadata for p in self.grads.params)

    def _graph_static_dataset(self):
        """The dataset of this epoch behind fixed addresses (a captured graph keeps reading the same buffers)."""
        if self._g_data is None:
            # the tensors of the first graphed epoch BECOME the persistent buffers (the dict keeps them alive): rollout-buffer views
            # keep their address from epoch to epoch and are never copied, per-epoch temporaries are copied into these
            self._g_data, seen = {}, set()
            for k, v in self.dataset.items():   # (two keys may share one tensor: the replay batch IS the agent batch while the buffer is empty)
                self._g_data[k] = v.clone() if v.data_ptr() in seen else v
                seen.add(v.data_ptr())
        for k, v in self.dataset.items():
            g = self._g_data[k]
            if g.shape != v.shape:
                raise RuntimeError("dataset shape changed under a captured update graph")
            if g.data_ptr() != v.data_ptr():
                g.copy_(v)
        return self._g_data

    def _graph_step_body(self):
        info = self._fwd_bwd({"_dataset": self._g_data, "_idx": self._g_idx, "_amp_idx": self._g_idx[:self._amp_minibatch_size]}, want_info=False)
        if info is None:     # raw vector of the fused kernels (see _raw_buffer): one launch
            self._g_keys = None
            if self._g_info.numel() != self._raw.numel():
                self._g_info = torch.zeros_like(self._raw)
            self._g_info += self._raw
            return
        self._g_keys = list(info)
        self._g_info += torch.stack([info[k].float().reshape(()) for k in self._g_keys])

    def _graph_tail(self, fuse_opt):
        """Behind the two passes of a branched step: the step's scalars into the epoch's accumulator, clip + Adam (one rank)."""
        self._g_keys = None
        if self._g_info.numel() != self._raw.numel():
            self._g_info = torch.zeros_like(self._raw)
        self._g_info += self._raw
        if fuse_opt:
            self._clip_and_step(step_device=self._g_step)

    def _graph_opt_key(self):
        """What a graph with the optimizer step inside has baked in: a change (checkpoint with another lr, ...) forces a re-capture."""
        g = self.optimizer.param_groups[0]
        collectives = self.multi_gpu or (self._force_collectives and self.dist is not None) or bool(os.environ.get("PHC_NO_OPT_IN_GRAPH"))
        return (not collectives, float(g["lr"]), tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"]), self.grad_norm if self.truncate_grads else None)

    def _graph_update(self):
        """All mini-epochs of one epoch through the captured forward / backward; returns the mean info dict (device tensors).
        One rank (no collective between backward and optimizer): the clip + Adam launches are part of the captured step as well -- a step is
        then the row-index copy and ONE replay, which takes the host out of the loop (on a freshly started box the eager launches between
        the replays cost ~10 % of the update: 69 vs 62 ms)."""
        self.set_train()
        self._graph_static_dataset()
        key = self._graph_opt_key()
        if self._graph is not None and getattr(self, "_graph_key", None) != key:
            self._graph = None
        fuse_opt = key[0]
        from .fast_ops import adam_state, _workspace
        adam_state(self.optimizer, self.grads.flat_param)      # exists before any capture
        _workspace("adam", L.load().phc_adam_workspace(), self.grads.flat_param.device, torch.float64)
        if self._g_idx is None:
            self._g_idx = torch.zeros(self.minibatch_size, dtype=torch.int64, device=self.device)
            self._g_info = torch.zeros(len(self._probe_info_keys()), dtype=torch.float32, device=self.device)
        if self._graph is None:
            if self._stale_grad_accumulators():
                raise RuntimeError("an autograd graph outside the update holds a parameter's gradient accumulator (e.g. `p.clone()` kept "
                                   "alive: use `p.detach().clone()`); stream capture would crash")
            self._g_idx.copy_(self._idx_buf[:self.minibatch_size])
            # the warm-up passes torch asks for before a capture must not train: they only touch the gradients (zeroed by every step)
            # and the normaliser statistics, which are restored afterwards
            norms = [(m, [b.clone() for b in m.buffers()]) for m in self._norms()]
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self._graph_step_body()
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                # with a process group alive its watchdog thread polls events while we capture: thread-local capture mode keeps
                # those calls from invalidating the capture (the graph itself holds no collective)
                in_group = self.dist is not None and self.dist.is_initialized()
                if in_group:   # nothing of the group may be in flight on this device while the capture starts
                    torch.cuda.synchronize()
                    self.dist.barrier()
                    torch.cuda.synchronize()
                if fuse_opt and getattr(self, "_g_step", None) is None:
                    self._g_step = torch.zeros((), dtype=torch.int64, device=self.device)
                mode = "thread_local" if in_group else "global"
                br = self._branch_streams(self.device) if (self._disc_grad_penalty > 0 and self._normalize_amp_input) else None
                if br:
                    # three LINEAR graphs per step: the policy pass, the discriminator pass captured ON its stream, and the tail (scalars,
                    # clip + Adam).  One graph with the passes as branches was measured first: a graph with a fork is enqueued node by node by
                    # the host (286 vs 40 us per replay in scripts/probes/graph_branch_concurrency.py) in topological order, and the device
                    # ran the chains mostly one after the other (profiles/r04_ppo/README.md)
                    gd, gt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    data, m = self._g_data, self._amp_minibatch_size
                    with torch.cuda.graph(g, capture_error_mode=mode):
                        self._policy_pass(data, self._g_idx)
                    with torch.cuda.graph(gd, stream=br, capture_error_mode=mode):
                        self._disc_pass(data, self._g_idx[:m])
                    with torch.cuda.graph(gt, capture_error_mode=mode):
                        self._graph_tail(fuse_opt)
                    g = (g, gd, gt)
                else:
                    with torch.cuda.graph(g, capture_error_mode=mode):
                        self._graph_step_body()
                        if fuse_opt:
                            self._clip_and_step(step_device=self._g_step)
            finally:
                for m, bufs in norms:
                    for b, k in zip(m.buffers(), bufs):
                        b.copy_(k)
            self._graph, self._graph_key = g, key
        self._g_info.zero_()
        st = self.optimizer.state[self.grads.flat_param]
        if fuse_opt:   # the device step count follows the optimizer's (checkpoint restores, eager steps in between)
            self._g_step.fill_(int(st["step"].item()))
        n = 0
        for _ in range(self.mini_epochs_num):
            for i in range(self.num_minibatches):
                s, e = i * self.minibatch_size, (i + 1) * self.minibatch_size
                self._g_idx.copy_(self._idx_buf[s:e])
                if e >= self.batch_size:
                    self._idx_buf[:] = torch.randperm(self.batch_size, device=self._idx_buf.device)
                if isinstance(self._graph, tuple):
                    gp, gd, gt = self._graph
                    main, sd = torch.cuda.current_stream(self.device), self._branches
                    split = (not fuse_opt) and self._split_active()
                    sd.wait_stream(main)
                    with torch.cuda.stream(sd):
                        gd.replay()
                        if split:
                            self._grad_all_reduce("disc")
                    gp.replay()
                    if split:
                        self._grad_all_reduce("policy")
                    main.wait_stream(sd)
                    gt.replay()
                else:
                    split = False
                    self._graph.replay()
                if not fuse_opt:
                    if not split:
                        self._grad_all_reduce()
                    self._clip_and_step()
                if self._trace is not None and self._g_keys is None:
                    inf = self._info_from_raw(self._raw.clone())
                    self._trace.append(torch.stack([inf[k] for k in self._probe_info_keys()]))
                n += 1
        if fuse_opt:
            st["step"] += n
        mean = self._g_info / n
        if self._g_keys is None:
            return self._info_from_raw(mean)
        return {k: mean[j] for j, k in enumerate(self._g_keys)}

    def _probe_info_keys(self):
        return ["actor_loss", "critic_loss", "b_loss", "entropy", "kl", "actor_clip_frac", "disc_loss", "disc_grad_penalty", "disc_logit_loss", "disc_agent_acc",
                "disc_demo_acc", "disc_agent_logit", "disc_demo_logit"]

    # ------------------------------------------------------------------ epoch (amp_agent.py:413-532)
    def _init_amp_demo_buf(self):
        n = int(np.ceil(self._amp_obs_demo_buffer.get_buffer_size() / self._amp_batch_size))
        for _ in range(n):
            self._amp_obs_demo_buffer.store({"amp_obs": self.vec_env.fetch_amp_obs_demo(self._amp_batch_size)})

    def _update_amp_demos(self):
        self._amp_obs_demo_buffer.store({"amp_obs": self.vec_env.fetch_amp_obs_demo(self._amp_batch_size)})

    def _store_replay_amp_obs(self, amp_obs):
        size = self._amp_replay_buffer.get_buffer_size()
        if self._amp_replay_buffer.get_total_count() > size:
            keep = torch.bernoulli(torch.full((amp_obs.shape[0],), self._amp_replay_keep_prob, device=amp_obs.device)) == 1.0
            amp_obs = amp_obs[keep]
        if amp_obs.shape[0] > size:
            amp_obs = amp_obs[torch.randperm(amp_obs.shape[0], device=amp_obs.device)[:size]]
        self._amp_replay_buffer.store({"amp_obs": amp_obs})

    def pre_epoch(self, epoch_num):
        t = self.task
        if (epoch_num > 1) and epoch_num % getattr(t, "shape_resampling_interval", 10 ** 9) == 1 and hasattr(t, "resample_motions"):
            t.resample_motions()
        if getattr(t, "getup_schedule", False) and hasattr(t, "update_getup_schedule"):  # amp_agent.py:518-525
            t.update_getup_schedule(epoch_num, getup_udpate_epoch=t.getup_udpate_epoch)
            warm = epoch_num > t.getup_udpate_epoch
            self._task_reward_w, self._disc_reward_w = (0.5, 0.5) if warm else (0, 1)
        self._snapshot_running_mean_std()

    def _snapshot_running_mean_std(self):
        """`self.running_mean_std_temp = copy.deepcopy(self.running_mean_std); .freeze()` (amp_agent.py:527-532), done as an
        in-place copy into ONE persistent module so that captured graphs keep valid buffer addresses."""
        if self.running_mean_std is None:
            return
        if self.running_mean_std_temp is None:
            self.running_mean_std_temp = copy.deepcopy(self.running_mean_std)
            self.running_mean_std_temp.freeze()
        else:
            for a, b in zip(self.running_mean_std_temp.buffers(), self.running_mean_std.buffers()):
                a.copy_(b)

    def post_epoch(self, epoch_num):
        self._snapshot_running_mean_std()
        for m in self._norms():
            m.sync(self.dist)

    def init_train(self):
        self.obs = self.env_reset()
        self._init_amp_demo_buf()

    def train_epoch(self):
        self.epoch_num += 1
        self.pre_epoch(self.epoch_num)
        sync = torch.cuda.synchronize if str(self.device).startswith("cuda") else (lambda: None)
        sync()
        t0 = time.time()
        with torch.no_grad(), self.grads.shadow_scope(), _marker("phc:rollout"):   # rollout inference reads the bf16 parameter copies
            batch = self.play_steps()
        sync()
        t1 = time.time()
        self._update_amp_demos()
        n = batch["amp_obs"].shape[0]
        batch["amp_obs_demo"] = self._amp_obs_demo_buffer.sample(n)["amp_obs"]
        batch["amp_obs_replay"] = batch["amp_obs"] if self._amp_replay_buffer.get_total_count() == 0 else self._amp_replay_buffer.sample(n)["amp_obs"]
        self.set_train()
        self.prepare_dataset(batch)
        infos, ginfo = [], None
        self._step_in_epoch = 0
        with self.grads.shadow_scope(), _marker("phc:update"):
            if self._graph_enabled():
                try:
                    ginfo = self._graph_update()
                except Exception as exc:   # capture not possible on this stack: eager launches from here on
                    if self._graph is not None:
                        raise
                    self._graph_failed = True
                    torch.cuda.synchronize()
                    print(f"[phc_amd] update graph disabled ({type(exc).__name__}: {exc})", flush=True)
            if ginfo is None:
                for _ in range(self.mini_epochs_num):
                    for i in range(self.num_minibatches):
                        infos.append(self.calc_gradients(self._get_item(i)))
                        if self._trace is not None:
                            self._trace.append(torch.stack([infos[-1][k].float().reshape(()) for k in self._probe_info_keys()]))
        self._store_replay_amp_obs(batch["amp_obs"])
        self.post_epoch(self.epoch_num)
        sync()
        t2 = time.time()
        self.frame += self.batch_size * self.world
        info = {k: v.item() for k, v in ginfo.items()} if ginfo is not None else {k: torch.stack([i[k] for i in infos]).mean().item() for k in infos[0]}
        # (one host transfer for the rollout's reward statistics: amp_agent.py:900-925 logs mean / std of the discriminator reward, the mean combined reward and the mean return)
        dr_std, dr_mean = torch.std_mean(batch["disc_rewards"])
        rs = torch.stack([batch["rewards"].mean(), dr_mean, dr_std, batch["mb_rewards"].mean(), batch["returns"].mean()]).tolist()
        info.update(play_time=t1 - t0, update_time=t2 - t1, total_time=t2 - t0, mean_task_reward=rs[0], mean_disc_reward=rs[1], disc_reward_std=rs[2],
                    mean_mb_reward=rs[3], mean_return=rs[4], reward_raw=batch["reward_raw"].tolist(),
                    step_fps=self.batch_size / (t1 - t0), total_fps=self.batch_size / (t2 - t0))  # common_agent.py:134-138
        if self._trace is not None:   # `+learning.params.config.trace_minibatches=True`: every optimizer step's scalars, in order (a diagnostic)
            tr = torch.stack(self._trace).cpu() if self._trace else torch.zeros(0)
            keys = self._probe_info_keys()
            info["minibatch_trace"] = {k: tr[:, j].tolist() for j, k in enumerate(keys)} if tr.numel() else {}
            self._trace = []
        return info

    def assemble_train_info(self, info):
        """The scalars the reference hands its SummaryWriter / wandb every epoch, under the reference's tags (`CommonAgent._assemble_train_info`,
        common_agent.py:603-626; `AMPAgent._assemble_train_info`, amp_agent.py:900-933): performance, learning rate, losses, discriminator statistics, reward
        terms; plus the evaluation sweep's `eval/*` entries when the epoch ran one (im_amp.py:334-346).  Pinned to the reference's method by tests/test_learner_epoch.py."""
        raw = list(info.get("reward_raw", [])) + [0.0] * 5
        out = {"performance/update_time": info["update_time"], "performance/play_time": info["play_time"], "performance/total_fps": info["total_fps"],
               "learning_rate/last_lr": self.last_lr, "learning_rate/lr_mul": 1.0, "learning_rate/e_clip": self.e_clip,
               "loss/actor_loss": info["actor_loss"], "loss/critic_loss": info["critic_loss"], "loss/bounds_loss": info["b_loss"], "loss/entropy": info["entropy"],
               "loss/kl": info["kl"], "loss/clip_frac": info["actor_clip_frac"], "disc/loss": info["disc_loss"], "disc/agent_acc": info["disc_agent_acc"],
               "disc/demo_acc": info["disc_demo_acc"], "disc/agent_logit": info["disc_agent_logit"], "disc/demo_logit": info["disc_demo_logit"],
               "disc/grad_penalty": info["disc_grad_penalty"], "disc/logit_loss": info["disc_logit_loss"], "disc/reward_mean": info["mean_disc_reward"],
               "disc/reward_std": info["disc_reward_std"], "rewards/returns": info["mean_return"],
               "rewards/mb_rewards": info["mean_mb_reward"], "rewards/body_pos": raw[0], "rewards/body_rot": raw[1], "rewards/lin_vel": raw[2], "rewards/ang_vel": raw[3],
               "rewards/power": raw[4]}
        out.update({k: v for k, v in info.items() if k.startswith("eval/")})
        return {k: float(v) for k, v in out.items()}

    def _log_train_info(self, scalars, output_dir):
        """`CommonAgent._log_train_info` (common_agent.py:627-635): `writer.add_scalar(tag, value, epoch)` for every entry.  tensorboardX / wandb are not part of
        this image; the same (tag, value, step) stream goes to `<output_dir>/summaries/scalars.jsonl`, one line per epoch, and -- where a `torch.utils.tensorboard`
        SummaryWriter can be made -- into event files next to it."""
        import json
        d = os.path.join(output_dir, "summaries")
        if getattr(self, "_summary_file", None) is None:
            os.makedirs(d, exist_ok=True)
            self._summary_file = open(os.path.join(d, "scalars.jsonl"), "a")
            try:
                from torch.utils.tensorboard import SummaryWriter
                self._summary_writer = SummaryWriter(d)
            except Exception:   # noqa: BLE001  (no tensorboard package: the jsonl stream is the record)
                self._summary_writer = None
        self._summary_file.write(json.dumps({"step": self.epoch_num, "frame": self.frame, **scalars}) + "\n")
        self._summary_file.flush()
        if self._summary_writer is not None:
            for k, v in scalars.items():
                self._summary_writer.add_scalar(k, v, self.epoch_num)

    def train(self, max_epochs, log=print, output_dir=None):
        """Training loop with the reference's checkpoint / evaluation cadence (common_agent.py:142-165): with `output_dir`,
        `Humanoid.pth` every min(50, save_best_after) epochs; every `save_frequency` epochs (save_intermediate) also
        `Humanoid_{epoch:08d}.pth` and the evaluation sweep `eval()`, which re-weights the clip sampling (auto-PMCP, im_amp.py:126-132)
        and writes `failed_{epoch:010d}.pkl`."""
        self.init_train()
        c = self.config
        save_freq, save_best_after = int(c.get("save_frequency", 0)), int(c.get("save_best_after", 100))
        save_intermediate = bool(c.get("save_intermediate", False))
        info = {}
        for _ in range(max_epochs):
            info = self.train_epoch()
            if self.rank == 0 and log is not None:
                log(f"epoch {self.epoch_num}: total_fps {info['total_fps']:.0f} step_fps {info['step_fps']:.0f} task_r {info['mean_task_reward']:.4f} "
                    f"disc_r {info['mean_disc_reward']:.4f} a_loss {info['actor_loss']:.4f} c_loss {info['critic_loss']:.4f} disc_loss {info['disc_loss']:.4f}")
            if output_dir is not None and save_freq > 0:
                if self.epoch_num % min(50, save_best_after) == 0 and self.rank == 0:
                    os.makedirs(output_dir, exist_ok=True)
                    self.save(os.path.join(output_dir, "Humanoid.pth"))
                if save_intermediate and self.epoch_num % save_freq == 0:
                    if self.rank == 0:
                        os.makedirs(output_dir, exist_ok=True)
                        self.save(os.path.join(output_dir, f"Humanoid_{self.epoch_num:08d}.pth"))
                    if hasattr(self.task, "_motion_lib") and hasattr(self.task, "_termination_distances"):
                        eval_info, _ = self.eval(output_dir=output_dir, log=log)
                        info.update(eval_info)
            if output_dir is not None and self.rank == 0:
                self._log_train_info(self.assemble_train_info(info), output_dir)
        return info

    def eval(self, output_dir=None, log=print):
        """IMAmpAgent.eval (im_amp.py:136-242): success rate / MPJPE over the whole motion set + auto-PMCP re-weighting."""
        from .im_eval import evaluate
        return evaluate(self, output_dir=output_dir, log=log if self.rank == 0 else None)

    # ------------------------------------------------------------------ checkpoint (amp_agent.py:69-108; SURVEY B4 key names)
    def _optimizer_state_dict(self):
        """The flat Adam state in the layout of the reference's `Adam(model.parameters())` (one entry per parameter, in
        `model.parameters()` order, common_agent.py:67): `exp_avg` / `exp_avg_sq` split at the bucket offsets.  Parameters that do
        not train (frozen PNN columns, the fixed sigma) appear in `params` without state, as torch writes them."""
        sd = self.optimizer.state_dict()
        flat_state = sd["state"].get(0, {})
        all_params = list(self.model.parameters())
        index = {id(p): i for i, p in enumerate(all_params)}
        state = {}
        for i, p in enumerate(self.grads.params):
            if flat_state:
                state[index[id(p)]] = {"step": flat_state["step"].clone() if torch.is_tensor(flat_state["step"]) else flat_state["step"],
                                       "exp_avg": self.grads.param_view(flat_state["exp_avg"], i).clone(),
                                       "exp_avg_sq": self.grads.param_view(flat_state["exp_avg_sq"], i).clone()}
        group = dict(sd["param_groups"][0])
        group["params"] = list(range(len(all_params)))
        return {"state": state, "param_groups": [group]}

    def _load_optimizer_state_dict(self, sd):
        """Inverse of `_optimizer_state_dict`; also accepts the one-entry flat layout of round-1 checkpoints.  A state that does not
        fit the current set of trainable parameters (another PNN stage, another network) is skipped with a warning, never mis-applied."""
        groups = sd.get("param_groups", [])
        n_entries = sum(len(g["params"]) for g in groups)
        if n_entries == 1 and len(list(self.model.parameters())) != 1:   # flat layout
            st = sd["state"].get(0)
            if st is not None and st["exp_avg"].numel() != self.grads.flat_param.numel():
                return self._skip_optimizer("flat optimizer state of another size")
            self.optimizer.load_state_dict(sd)
            return True
        all_params = list(self.model.parameters())
        if n_entries != len(all_params):
            return self._skip_optimizer(f"optimizer state for {n_entries} parameters, the model has {len(all_params)}")
        index = {id(p): i for i, p in enumerate(all_params)}
        state = sd["state"]
        mine = [index[id(p)] for p in self.grads.params]
        have = [i for i in mine if i in state]
        if not state:         # a checkpoint saved before the first optimizer step
            return True
        if len(have) != len(mine) or any(i not in mine for i in state):
            return self._skip_optimizer("optimizer state belongs to another set of trainable parameters (e.g. another PNN stage)")
        for p, i in zip(self.grads.params, mine):
            if tuple(state[i]["exp_avg"].shape) != tuple(p.shape):
                return self._skip_optimizer(f"optimizer state shape mismatch at parameter {i}")
        dev = self.grads.flat_param.device
        step = state[mine[0]]["step"]
        flat = {"step": (step.clone().float() if torch.is_tensor(step) else torch.tensor(float(step))),
                "exp_avg": torch.zeros_like(self.grads.flat_param), "exp_avg_sq": torch.zeros_like(self.grads.flat_param)}
        for k, i in enumerate(mine):   # (K-padded weights: the moments of the pad elements are zero)
            for key in ("exp_avg", "exp_avg_sq"):
                self.grads.param_view(flat[key], k).copy_(state[i][key].to(dev).float())
        group = {k: v for k, v in groups[0].items() if k != "params"}
        cur = self.optimizer.state_dict()["param_groups"][0]
        merged = dict(cur)
        merged.update({k: v for k, v in group.items() if k in cur})
        merged["params"] = [0]
        self.optimizer.load_state_dict({"state": {0: flat}, "param_groups": [merged]})
        return True

    def _skip_optimizer(self, why):
        if self.rank == 0:
            print(f"[phc_amd] optimizer state not restored: {why}", flush=True)
        return False

    def get_full_state_weights(self):
        s = {"model": self.model.state_dict(), "epoch": self.epoch_num, "optimizer": self._optimizer_state_dict(), "frame": self.frame}
        if self.running_mean_std is not None:
            s["running_mean_std"] = self.running_mean_std.state_dict()
        if self.value_mean_std is not None:
            s["reward_mean_std"] = self.value_mean_std.state_dict()
        if self._amp_input_mean_std is not None:
            s["amp_input_mean_std"] = self._amp_input_mean_std.state_dict()
        if hasattr(self.task, "get_env_rng_state"):   # (an extra key: the reference's loaders read the keys above by name)
            s["env_state"] = self.task.get_env_rng_state()
        return s

    def set_full_state_weights(self, w, load_optimizer=True):
        self.model.load_state_dict(w["model"])
        self.epoch_num = w.get("epoch", 0)
        self.frame = w.get("frame", 0)
        if "optimizer" in w and load_optimizer:
            self._load_optimizer_state_dict(w["optimizer"])
        for key, mod in (("running_mean_std", self.running_mean_std), ("reward_mean_std", self.value_mean_std), ("amp_input_mean_std", self._amp_input_mean_std)):
            if mod is not None and key in w:
                mod.load_state_dict(w[key])
        if self.grads.shadow is not None:
            self.grads.shadow.copy_(self.grads.flat_param)
        if "env_state" in w and hasattr(self.task, "set_env_rng_state"):
            self.task.set_env_rng_state(w["env_state"])

    def save(self, path):
        torch.save(self.get_full_state_weights(), path)

    def restore(self, path, load_optimizer=True):
        """`IMAmpAgent.restore` (im_amp.py:101-117): the checkpoint, then the newest `failed_*.pkl` next to it -- the termination
        history of the last evaluation sweep -- back into the motion library's sampling probabilities.  Players (`test=True`) pass
        `load_optimizer=False`, as the reference's `set_weights` path (amp_agent.py:84-90) does not touch the optimizer."""
        self.set_full_state_weights(torch.load(path, map_location=self.device, weights_only=False), load_optimizer=load_optimizer)
        import glob
        fails = glob.glob(os.path.join(os.path.dirname(os.path.abspath(path)), "failed_*"))
        lib = getattr(self.task, "_motion_lib", None)
        if fails and lib is not None and hasattr(lib, "update_sampling_prob"):
            import joblib
            newest = sorted(fails, key=lambda x: int(x.split("_")[-1].split(".")[0]))[-1]
            hist = joblib.load(newest)["termination_history"]
            ok = lib.update_sampling_prob(torch.as_tensor(hist, dtype=lib._termination_history.dtype, device=lib._termination_history.device))
            if self.rank == 0:
                print(f"[phc_amd] termination history {'restored from' if ok else 'does not match the motion set:'} {newest}", flush=True)
