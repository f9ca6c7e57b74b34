"""Config objects: nested attribute bags with the reference's field names.

The reference declares configs as nested Python classes that `BaseConfig.__init__` instantiates
recursively (airgym/envs/base/base_config.py:33-55) so that `cfg.env.num_envs` is a per-instance attribute
`update_cfg_from_args` can overwrite.  Here the same objects are produced from a plain dict spec:
`make_config_class(name, spec)` returns a class whose nested sections are real (inner) classes - user code
that subclasses a section or reads `Cfg.env.num_envs` at class level keeps working - and whose instances
get fresh section instances.
"""
import copy
import inspect


class BaseConfig:
    def __init__(self) -> None:
        self.init_member_classes(self)

    @staticmethod
    def init_member_classes(obj):
        for key in dir(obj):
            if key == "__class__":
                continue
            member = getattr(obj, key)
            if inspect.isclass(member):
                inst = member()
                setattr(obj, key, inst)
                BaseConfig.init_member_classes(inst)


class Section:
    """Marks a dict in a spec as a nested config section (anything else is a leaf value)."""

    def __init__(self, **fields):
        self.fields = fields


def _build(name, fields, bases=(object,)):
    ns = {}
    for k, v in fields.items():
        ns[k] = _build(k, v.fields) if isinstance(v, Section) else copy.deepcopy(v)
    return type(name, bases, ns)


def make_config_class(name, spec, doc=""):
    cls = _build(name, spec, (BaseConfig,))
    cls.__doc__ = doc
    return cls
