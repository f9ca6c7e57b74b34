"""Nested-class configs (same mechanism as the reference's airgym/envs/base/base_config.py:33-55):
every inner class of a config class is instantiated recursively so `cfg.env.num_envs` is an instance
attribute that `update_cfg_from_args` may overwrite."""
import inspect


class BaseConfig:
    def __init__(self) -> None:
        self.init_member_classes(self)

    @staticmethod
    def init_member_classes(obj):
        for key in dir(obj):
            if key == "__class__":
                continue
            var = getattr(obj, key)
            if inspect.isclass(var):
                inst = var()
                setattr(obj, key, inst)
                BaseConfig.init_member_classes(inst)
