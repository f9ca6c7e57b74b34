"""The tail of a single-GPU optimizer step as ONE launch (csrc/update_tail.hip: second reduction stage, gradient norm, clip + Adam +
KL-adaptive LR - trancate_gradients_and_step, lib/agent/a2c_base.py:293-316; schedulers.py:19-32 - and the next step's bf16 weight
images behind grid barriers) against the five separate launches it replaces.  Every phase runs the device body of the launch it
replaces (csrc/tail_parts.hpp), so the two runs must be the SAME run bit for bit: parameters, Adam moments, {lr, step}, the clipped
gradient, the weight images, the logged losses - eager and as minibatch hipGraphs."""
import os

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params(fuse_tail, graph_update, envs=4096, minibatches=8):
    with open(os.path.join(REPO, "scripts", "config", "ppo_hovering.yaml")) as f:
        params = yaml.safe_load(f)["params"]
    c = params["config"]
    params["network"]["mlp"]["units"] = [256, 256]
    c.update(num_actors=envs, minibatch_size=envs * c["horizon_length"] // minibatches, device="cuda:0", multi_gpu=False,
             max_epochs=-1, write_summaries=False, print_stats=False, save_frequency=0, save_best_after=10 ** 9,
             use_hip_graph=True, use_hip_graph_update=graph_update, fuse_update_tail=fuse_tail)
    c["env_config"] = {"use_image": False, "num_envs": envs, "ctl_mode": "rate", "seed": 0, "sim_device": "cuda:0", "headless": True}
    params["seed"] = 0
    return params


def _run(fuse_tail, graph_update, epochs=4):
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent
    torch.manual_seed(0)
    agent = A2CAgent("tail", _params(fuse_tail, graph_update))
    agent.init_tensors()
    agent.obs = agent.env_reset()
    stats = []
    for _ in range(epochs):
        agent.epoch_num += 1
        st = agent.train_epoch()
        stats.append((st["kl"], st["a_loss"], st["c_loss"], st["last_lr"]))
    fs = agent._fused_step
    sg = fs.split[len(fs.layers) - 1]
    out = {"param": agent.flat_param.clone(), "grad": agent.flat_grad.clone(), "m": agent.optimizer.exp_avg.clone(),
           "v": agent.optimizer.exp_avg_sq.clone(), "state": agent.optimizer.state[:2].clone(), "stats": stats,
           "image": sg.in_image.clone(), "bwd": sg.bwd.clone(), "fusable": fs.tail_fusable(), "fresh": fs.images_fresh,
           "graphs": len(agent._upd_graphs), "rms": agent.model.running_mean_std.running_mean.clone()}
    agent.vec_env.env.hip.close()
    return out


@pytest.mark.parametrize("graph_update", [False, True])
def test_one_launch_tail_is_the_five_launch_tail_bit_for_bit(graph_update):
    assert torch.cuda.is_available()
    sep = _run(False, graph_update)
    one = _run(True, graph_update)
    assert one["fusable"] and not sep["fusable"] and one["fresh"] and not sep["fresh"]
    assert (one["graphs"] > 0) == graph_update
    for k in ("param", "grad", "m", "v", "state", "rms"):
        assert torch.equal(one[k], sep[k]), k
    assert one["stats"] == sep["stats"]
    # the images the fused tail left behind are the images of the FINAL parameters (what the next step would otherwise prepare);
    # the five-launch run's images are one optimizer step older, so compare against a fresh preparation instead
    import ctypes

    from airgym_amd import _native as N
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent
    lib = N.load()
    agent = A2CAgent("tail_ref", _params(False, False))
    with torch.no_grad():
        agent.flat_param.copy_(one["param"])
    fs = agent._fused_step
    sg = fs.split[len(fs.layers) - 1]
    sg.prepare_input_image(fs.layers[0][0], fs.layers[0][1])
    torch.cuda.synchronize()
    assert torch.equal(sg.in_image, one["image"]) and torch.equal(sg.bwd, one["bwd"])
    agent.vec_env.env.hip.close()


def test_tail_entry_point_refuses_what_it_cannot_run():
    from airgym_amd import _native as N
    lib = N.load()
    assert lib.ag_update_tail_barrier_bytes() >= 4
    z = torch.zeros(64, device="cuda")
    # NULL barrier / NULL parameter buffer: AG_ERR_INVALID_ARG (-1), nothing launched
    jobs = (N.AgSumJob * 1)(N.AgSumJob(z.data_ptr(), z.data_ptr(), 8, 8))
    st = torch.zeros(40, dtype=torch.float64, device="cuda")
    args = [jobs, 1, z.data_ptr(), 64, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), st.data_ptr(), 8,
            0.9, 0.999, 1e-8, 0.0, 1.0, 0.008, 1e-6, 1e-2, z.data_ptr(), z.data_ptr(), 18, z.data_ptr(), z.data_ptr(), None]
    assert lib.ag_update_tail(*args, None, None) == -1
    bar = torch.zeros(16, dtype=torch.int32, device="cuda")
    bad = list(args); bad[20] = 24                      # an input width the fused first layer does not exist for
    assert lib.ag_update_tail(*bad, bar.data_ptr(), None) == -6
