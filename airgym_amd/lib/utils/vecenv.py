"""VecEnv adapter - drop-in boundary #1 (reference: lib/utils/vecenv.py:13-119).

`create_vec_env(name, num_actors, **env_config)` -> AirGymRLGPUEnv whose `step` returns
(obs, rew, done, info) with the privileged observations stripped, `reset` returns obs, and
`get_env_info` returns the cfg.env class attributes plus action/observation spaces.
"""
from argparse import Namespace

import numpy as np

import airgym_amd.envs  # noqa: F401  (registers the tasks)
from airgym_amd.lib.utils import env_configurations
from airgym_amd.lib.utils.ivecenv import IVecEnv
from airgym_amd.lib.utils.spaces import Box, Dict
from airgym_amd.utils.task_registry import task_registry

vecenv_config = {}


def register(config_name, func):
    vecenv_config[config_name] = func


def create_vec_env(config_name, num_actors, **kwargs):
    vec_env_name = env_configurations.configurations[config_name]["vecenv_type"]
    return vecenv_config[vec_env_name](config_name, num_actors, **kwargs)


def get_class_attributes(obj):
    cls = obj if isinstance(obj, type) else obj.__class__
    return {k: v for k, v in cls.__dict__.items() if not k.startswith("__") and not callable(v)}


class AirGymRLGPUEnv(IVecEnv):
    def __init__(self, config_name, num_actors, **kwargs):
        self.use_image = kwargs.get("use_image", False)
        kwargs.setdefault("num_envs", num_actors)
        self.env, self.env_info = env_configurations.configurations[config_name]["env_creator"](**kwargs)

    def step(self, actions):
        obs, _privileged, rewards, dones, infos = self.env.step(actions)   # ExtractObsWrapper, vecenv.py:50-67
        return obs, rewards, dones, infos

    def reset(self):
        obs, _privileged = self.env.reset()
        return obs

    def get_number_of_agents(self):
        return 1

    def get_env_info(self):
        info = get_class_attributes(self.env_info.env)
        info.update({k: v for k, v in vars(self.env_info.env).items() if not k.startswith("__")})
        info["action_space"] = Box(np.ones(self.env.num_actions) * -1.0, np.ones(self.env.num_actions) * 1.0)
        obs_box = Box(np.ones(self.env.num_obs) * -np.inf, np.ones(self.env.num_obs) * np.inf)
        if self.use_image:      # vecenv.py:93-98
            info["observation_space"] = Dict({
                "image": Box(0, 1, shape=(self.env.cam_channel, self.env.cam_resolution[0], self.env.cam_resolution[1])),
                "observation": obs_box})
        else:
            info["observation_space"] = obs_box
        return info


for _task_name in task_registry.get_registered_tasks():
    env_configurations.register(_task_name, {
        "env_creator": lambda task_name=_task_name, **kwargs: task_registry.make_env(task_name, args=Namespace(**kwargs)),
        "vecenv_type": "AirGym-RLGPU",
    })

register("AirGym-RLGPU", lambda config_name, num_actors, **kwargs: AirGymRLGPUEnv(config_name, num_actors, **kwargs))
