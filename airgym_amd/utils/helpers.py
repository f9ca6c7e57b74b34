"""CLI / config helpers with the reference's flag names (airgym/utils/helpers.py:23-116 and the
fallback argument parser airgym/utils/gym_utils/gymutil.py:298-366)."""
import argparse
from types import SimpleNamespace

SIM_PHYSX = "physx"   # stand-in for gymapi.SIM_PHYSX; the HIP integrator is the only engine


def class_to_dict(obj) -> dict:
    if not hasattr(obj, "__dict__"):
        return obj
    result = {}
    for key in dir(obj):
        if key.startswith("_"):
            continue
        val = getattr(obj, key)
        if isinstance(val, list):
            result[key] = [class_to_dict(item) for item in val]
        else:
            result[key] = class_to_dict(val)
    return result


def parse_sim_params(args, cfg):
    """gymapi.SimParams stand-in: an attribute bag with dt / substeps / gravity / use_gpu_pipeline
    (helpers.py:40-62).  Only dt reaches the kernel; gravity is fixed at (0,0,-9.81) like every shipped config."""
    sim = dict(cfg.get("sim", {}))
    p = SimpleNamespace(dt=sim.get("dt", 0.01), substeps=sim.get("substeps", 1),
                        gravity=sim.get("gravity", [0.0, 0.0, -9.81]), up_axis=sim.get("up_axis", 1),
                        use_gpu_pipeline=getattr(args, "use_gpu_pipeline", True),
                        physx=SimpleNamespace(**sim.get("physx", {})))
    g = list(p.gravity)
    if abs(g[0]) > 0 or abs(g[1]) > 0 or abs(g[2] + 9.81) > 1e-9:
        raise ValueError(f"gravity {g} is not supported: the kernel integrates with (0, 0, -9.81)")
    if p.substeps != 1:
        raise ValueError("substeps != 1 is not supported")
    return p


def update_cfg_from_args(env_cfg, args):
    """helpers.py:64-80: num_envs, ctl_mode, seed."""
    if env_cfg is not None:
        if getattr(args, "num_envs", None) is not None:
            env_cfg.env.num_envs = args.num_envs
        if hasattr(args, "ctl_mode"):
            env_cfg.env.ctl_mode = args.ctl_mode
        if hasattr(args, "seed"):
            env_cfg.seed = args.seed
        if getattr(args, "env_id_offset", None) is not None:
            env_cfg.env.env_id_offset = args.env_id_offset
        # this build's opt-ins.  The registered cfg object is shared between make_env calls, so each call starts from the
        # value the config CLASS declares (a task config may turn an opt-in on) and the command line only overrides it
        # when the flag was actually given.
        for key in ("fix_time_outs", "stagger_episode_phase"):
            default = bool(getattr(type(env_cfg.env), key, False))
            given = getattr(args, key, None)
            setattr(env_cfg.env, key, default if given is None else bool(given))
    return env_cfg


def get_args(argv=None):
    """Same flags as helpers.py:82-116 + gymutil.parse_arguments' device flags."""
    p = argparse.ArgumentParser(description="RL Policy")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--tf", action="store_true")
    p.add_argument("--train", action="store_true")
    p.add_argument("--play", action="store_true")
    p.add_argument("--checkpoint", type=str)
    p.add_argument("--num_envs", type=int, default=4096)
    p.add_argument("--sigma", type=float)
    p.add_argument("--track", action="store_true")
    p.add_argument("--wandb-project-name", type=str, default="rl_games")
    p.add_argument("--wandb-entity", type=str, default=None)
    p.add_argument("--task", type=str, default=None)
    p.add_argument("--experiment_name", type=str)
    p.add_argument("--headless", action="store_true", default=False)
    p.add_argument("--horovod", action="store_true", default=False)
    p.add_argument("--rl_device", type=str, default="cuda:0")
    p.add_argument("--ctl_mode", required=True, type=str,
                   help="Specify the control mode and the options are: pos, vel, atti, rate, prop")
    # isaacgym gymutil flags kept for command-line compatibility
    p.add_argument("--sim_device", type=str, default="cuda:0")
    p.add_argument("--pipeline", type=str, default="gpu")
    p.add_argument("--graphics_device_id", type=int, default=0)
    p.add_argument("--physx", action="store_true")
    p.add_argument("--flex", action="store_true")
    p.add_argument("--num_threads", type=int, default=0)
    p.add_argument("--subscenes", type=int, default=0)
    p.add_argument("--slices", type=int, default=None)
    # this build's opt-ins (default None = "not given": the task config's value stands, update_cfg_from_args)
    p.add_argument("--fix_time_outs", action="store_true", default=None,
                   help="extras['time_outs'] flags the envs that reached the time limit (the reference's never fires)")
    p.add_argument("--stagger_episode_phase", action="store_true", default=None,
                   help="Hovering: a full reset starts every env at its own progress (desynchronised time limits)")
    args = p.parse_args(argv)
    args.physics_engine = SIM_PHYSX
    args.use_gpu = True
    args.use_gpu_pipeline = args.pipeline.lower() in ("gpu", "cuda")
    dev = args.sim_device
    args.sim_device_type = "cuda" if dev.startswith("cuda") else dev
    args.compute_device_id = int(dev.split(":")[1]) if ":" in dev else 0
    args.sim_device_id = args.compute_device_id
    args.sim_device = f"cuda:{args.sim_device_id}" if args.sim_device_type == "cuda" else dev
    return args
