"""Runner: YAML -> agent -> train (reference: lib/torch_runner.py:10-100)."""
import os
import random

import numpy as np
import torch

from airgym_amd.lib.agent.a2c_continuous import A2CAgent


def _restore(agent, args):
    if args.get("checkpoint"):
        agent.restore(args["checkpoint"])


class Runner:
    def __init__(self, algo_observer=None):
        self.algo_observer = algo_observer
        self.algo_factory = {"a2c_continuous": A2CAgent}

    def reset(self):
        pass

    def load_config(self, params):
        """torch_runner.py:26-73: seed handling (per-rank offset), algo/model names, observer."""
        self.seed = params.get("seed", None)
        if self.seed is None:
            self.seed = int.from_bytes(os.urandom(3), "little")
        self.local_rank = self.global_rank = 0
        self.world_size = 1
        if params["config"].get("multi_gpu", False):
            self.local_rank = int(os.getenv("LOCAL_RANK", "0"))
            self.global_rank = int(os.getenv("RANK", "0"))
            self.world_size = int(os.getenv("WORLD_SIZE", "1"))
        # policy-sampling RNG differs per rank; the ENV RNG is keyed by global env id instead, so the
        # env seed stays the same on every rank (the reference offsets both: torch_runner.py:44,66)
        torch.manual_seed(self.seed + self.global_rank)
        np.random.seed(self.seed + self.global_rank)
        random.seed(self.seed + self.global_rank)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(self.seed + self.global_rank)
        params["config"].setdefault("env_config", {})
        params["config"]["env_config"].setdefault("seed", self.seed)
        self.algo_params = params["algo"]
        self.algo_name = self.algo_params["name"]
        self.exp_config = None
        params["config"].setdefault("features", {})
        params["config"]["features"]["observer"] = self.algo_observer
        self.params = params

    def load(self, yaml_conf):
        config = yaml_conf["params"]
        self.load_config(params=config)

    def run_train(self, args):
        print("Started to train")
        if self.algo_name not in self.algo_factory:
            raise ValueError(f"unknown algo {self.algo_name!r}")
        agent = self.algo_factory[self.algo_name]("run", self.params)
        _restore(agent, args)
        return agent.train()

    def run_play(self, args):
        print("Started to play")
        from airgym_amd.lib.agent.players import A2CPlayer
        player = A2CPlayer(self.params)
        if args.get("checkpoint"):
            player.restore(args["checkpoint"])
        return player.run()

    def run(self, args):
        """torch_runner.py:95-101: --train trains, --play plays, NEITHER flag trains (both are store_true
        flags, so the documented `runner.py --task hovering --ctl_mode rate --headless` has train=False)."""
        if args.get("train"):
            return self.run_train(args)
        if args.get("play"):
            return self.run_play(args)
        return self.run_train(args)
