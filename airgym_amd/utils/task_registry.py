"""TaskRegistry.make_env - the upper drop-in boundary (reference: airgym/utils/task_registry.py:35-112)."""
import os

import numpy as np
import torch

from airgym_amd.utils.helpers import class_to_dict, get_args, parse_sim_params, update_cfg_from_args


class TaskRegistry:
    def __init__(self):
        self.task_classes = {}
        self.env_cfgs = {}

    def register(self, name: str, task_class, env_cfg):
        self.task_classes[name] = task_class
        self.env_cfgs[name] = env_cfg

    def get_task_class(self, name: str):
        return self.task_classes[name]

    def get_cfgs(self, name):
        return self.env_cfgs[name]

    def get_registered_tasks(self):
        return list(self.task_classes.keys())

    def make_env(self, name, args=None, env_cfg=None):
        """-> (env, env_cfg).  Raises ValueError for an unregistered task (task_registry.py:78-79)."""
        if args is None:
            args = get_args()
        if name not in self.task_classes:
            raise ValueError(f"Task with name: {name} was not registered")
        task_class = self.get_task_class(name)
        if env_cfg is None:
            env_cfg = self.get_cfgs(name)
        env_cfg = update_cfg_from_args(env_cfg, args)

        seed = env_cfg.seed
        if seed is None or seed == -1:
            seed = np.random.randint(0, 10000)
            print("Setting seed: {}".format(seed))
            env_cfg.seed = seed
        np.random.seed(seed)
        torch.manual_seed(seed)
        os.environ["PYTHONHASHSEED"] = str(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)

        sim_params = parse_sim_params(args, {"sim": class_to_dict(env_cfg.sim)})
        env = task_class(cfg=env_cfg, sim_params=sim_params,
                         physics_engine=getattr(args, "physics_engine", None),
                         sim_device=getattr(args, "sim_device", "cuda:0"),
                         headless=getattr(args, "headless", True))
        return env, env_cfg


task_registry = TaskRegistry()
