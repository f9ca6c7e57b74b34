"""IVecEnv - the interface the PPO agent programs against (reference: lib/utils/ivecenv.py:1-36)."""


class IVecEnv:
    def step(self, actions):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def has_action_masks(self):
        return False

    def get_number_of_agents(self):
        return 1

    def get_env_info(self):
        pass

    def seed(self, seed):
        pass

    def set_train_info(self, env_frames, *args, **kwargs):
        """algo -> env information channel (curricula); unused by the shipped tasks."""
        pass

    def get_env_state(self):
        """Serializable env state for checkpoints; the reference always returns None (ivecenv.py:28-33)."""
        return None

    def set_env_state(self, env_state):
        pass
