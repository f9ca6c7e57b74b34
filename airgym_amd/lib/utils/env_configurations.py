"""name -> {env_creator, vecenv_type} table (reference: lib/utils/env_configurations.py)."""
configurations = {}


def register(name, config):
    configurations[name] = config


def get_env_info(env):
    result_shapes = {"observation_space": env.observation_space, "action_space": env.action_space,
                     "agents": 1, "value_size": 1}
    if hasattr(env, "get_number_of_agents"):
        result_shapes["agents"] = env.get_number_of_agents()
    return result_shapes
