"""Child process of test_update_graph_equals_eager_launches: train three epochs with the captured update graphs (policy pass and discriminator
pass side by side on two HIP streams), with eager launches, and with ONE captured graph on one stream, from the same seed; print one JSON line with
the differences."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phc_amd.config import compose  # noqa: E402
from phc_amd.env.tasks.vec_task import parse_task  # noqa: E402
from phc_amd.learning.amp_agent import IMAmpAgent  # noqa: E402


def run(graph, streams=True):
    torch.manual_seed(0)
    cfg = compose(["env.num_envs=256", "env.motion_file=synthetic:2:3", "learning.params.config.minibatch_size=2048",
                   "learning.params.config.amp_minibatch_size=1024", "learning.params.config.amp_obs_demo_buffer_size=4096",
                   "learning.params.config.amp_replay_buffer_size=4096", f"+learning.params.config.hip_graph={graph}", f"+learning.params.config.branch_streams={streams}"])
    task, env = parse_task(cfg)
    torch.manual_seed(11)
    agent = IMAmpAgent(env, cfg)
    p0 = agent.grads.flat_param.clone()
    agent.init_train()
    infos = [agent.train_epoch() for _ in range(3)]
    st = agent.optimizer.state[agent.grads.flat_param]
    return dict(graph=agent._graph is not None, three_graphs=isinstance(agent._graph, tuple), two_streams=agent._branches is not None, p0=p0, p=agent.grads.flat_param.clone(), mean=agent.running_mean_std.running_mean.clone(),
                var=agent._amp_input_mean_std.running_var.clone(), count=float(agent.running_mean_std.count), step=int(st["step"]),
                info=infos[-1], expected=3 * agent.mini_epochs_num * agent.num_minibatches)


a, b, c = run(True), run(False), run(True, streams=False)
print(json.dumps({
    "three_graphs": [a["three_graphs"], c["three_graphs"]], "two_streams": [a["two_streams"], b["two_streams"], c["two_streams"]],
    "param_maxdiff_one_stream": float((a["p"] - c["p"]).abs().max()), "count_equal_one_stream": a["count"] == c["count"],
    "graph_used": [a["graph"], b["graph"]], "steps": [a["step"], b["step"]], "expected_steps": a["expected"],
    "count_equal": a["count"] == b["count"], "mean_maxdiff": float((a["mean"] - b["mean"]).abs().max()),
    "var_relmaxdiff": float(((a["var"] - b["var"]).abs() / b["var"].abs().clamp_min(1e-12)).max()),
    "param_maxdiff": float((a["p"] - b["p"]).abs().max()), "param_update_size": float((b["p"] - b["p0"]).abs().max()),
    "info": {k: [a["info"][k], b["info"][k]] for k in ("actor_loss", "critic_loss", "disc_loss", "kl")}}))
