#!/usr/bin/env python3
"""Entry point with the reference's command line (scripts/runner.py:47-70):

    python scripts/runner.py --task hovering --ctl_mode rate --headless [--num_envs N --seed S --checkpoint P]

Multi-GPU (one process per GPU, RCCL):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/runner.py \
        --task hovering --ctl_mode rate --headless --multi_gpu
"""
import os
import sys

import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from airgym_amd.lib.torch_runner import Runner  # noqa: E402
from airgym_amd.utils.helpers import get_args  # noqa: E402


def update_config(config, args):
    """Merge the CLI into params.config.env_config (runner.py:19-44)."""
    c = config["params"]["config"]
    if args["task"] is not None:
        c["env_name"] = args["task"]
    if args["experiment_name"] is not None:
        c["name"] = args["experiment_name"]
    ec = c.setdefault("env_config", {})
    ec["headless"] = args["headless"]
    ec["num_envs"] = args["num_envs"]
    c["num_actors"] = args["num_envs"]
    ec["ctl_mode"] = args["ctl_mode"]
    ec["sim_device"] = args["sim_device"]
    ec["physics_engine"] = args["physics_engine"]
    ec["use_gpu"] = args["use_gpu"]
    ec["use_gpu_pipeline"] = args["use_gpu_pipeline"]
    ec["subscenes"] = args["subscenes"]
    ec["num_threads"] = args["num_threads"]
    if args["seed"] > 0:
        config["params"]["seed"] = args["seed"]
        ec["seed"] = args["seed"]
    c["device"] = args["rl_device"]
    if args.get("multi_gpu"):
        # one process per GPU: every rank simulates and learns on ITS device (the CLI default cuda:0 would put
        # every rank's env on GPU 0 while the agent lives on cuda:LOCAL_RANK)
        c["multi_gpu"] = True
        dev = "cuda:" + os.getenv("LOCAL_RANK", "0")
        ec["sim_device"] = dev
        c["device"] = dev
    return config


if __name__ == "__main__":
    argv = sys.argv[1:]
    multi_gpu = "--multi_gpu" in argv
    argv = [a for a in argv if a != "--multi_gpu"]
    args = vars(get_args(argv))
    args["multi_gpu"] = multi_gpu
    config_name = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config", "ppo_" + args["task"] + ".yaml")
    print("Loading config: ", config_name)
    with open(config_name, "r") as stream:
        config = yaml.safe_load(stream)
    config = update_config(config, args)
    # minibatch stays a fixed fraction of the batch when --num_envs changes (shipped: 4096*24/2048 = 48)
    c = config["params"]["config"]
    if c["num_actors"] * c["horizon_length"] % c["minibatch_size"] != 0:
        c["minibatch_size"] = c["num_actors"] * c["horizon_length"] // 8
    runner = Runner()
    runner.load(config)
    runner.reset()
    runner.run(args)
