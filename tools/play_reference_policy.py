#!/usr/bin/env python3
"""Behavioural cross-check of the parts of the env that cannot be pinned (integrator, PX4-style cascades, depth renderer):
fly the reference's own trained Planning policy (trained/planning_cnn_rate.pth, trained in IsaacGym + rlPx4Controller) in THIS
env and compare it with a random and a zero-action policy.  The checkpoint is not part of the repo: pass --checkpoint."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--image-flips", nargs="*", default=[], choices=["u", "v", "uv"],
                    help="renderer-convention probe: re-fly the reference policy with the image mirrored left-right (u), "
                         "upside-down (v) or both")
    ap.add_argument("--only-reference", action="store_true", help="skip the random / zero-action baselines")
    ap.add_argument("--thrust-gains", type=float, nargs="*", default=[],
                    help="system-identification probe: re-fly the reference policy with its thrust command scaled by k")
    args = ap.parse_args()
    from airgym_amd.lib.model.a2c_continuous_logstd_model import ModelA2CContinuousLogStd
    from airgym_amd.lib.utils import vecenv
    import airgym_amd.envs  # noqa: F401  (registers the tasks)
    params = {"network": {"separate": False, "mlp": {"units": [64, 128, 64], "activation": "elu"},
                          "space": {"continuous": {"fixed_sigma": True}}, "cnn": {"output_dim": 30}},
              "config": {"normalize_input": True, "normalize_value": True}}
    keys = {"actions_num": 4, "input_shape": {"image": (1, 212, 120), "observation": (16,)}}
    model = ModelA2CContinuousLogStd(params, keys)
    model.load_state_dict(torch.load(args.checkpoint, map_location="cpu", weights_only=False)["model"], strict=True)
    model = model.cuda().eval()
    out = {}
    names = ["reference_policy", "random", "zero"] + [f"reference_policy_thrust_x{k}" for k in args.thrust_gains] \
        + [f"reference_policy_flip_{f}" for f in args.image_flips]
    if args.only_reference:
        names = [n for n in names if n.startswith("reference_policy")]
    for name in names:
        env = vecenv.create_vec_env("planning", args.envs, use_image=True, num_envs=args.envs, ctl_mode="rate", seed=0,
                                    sim_device="cuda:0", headless=True)
        obs = env.reset()
        g = torch.Generator(device="cuda").manual_seed(1)
        ep_len = torch.zeros(args.envs, device="cuda")
        ep_rew = torch.zeros(args.envs, device="cuda")
        done_len, done_rew, n_done, rew_sum, n_goal = 0.0, 0.0, 0, 0.0, 0
        causes = {"altitude_band": 0, "heading": 0, "collision_or_bounds": 0}
        thrust_sum, speed_sum = 0.0, 0.0
        lat_sum, up_sum, act_sum = 0.0, 0.0, torch.zeros(4, device="cuda")
        for t in range(args.steps):
            with torch.no_grad():
                if name.startswith("reference_policy"):
                    image = obs["image"]
                    if "_flip_" in name:
                        f = name.split("_flip_")[1]
                        image = image.flip(dims=[d for d, c in ((2, "u"), (3, "v")) if c in f])
                    mu, _, _ = model.trunk({"image": image, "observation": obs["observation"]})
                    act = mu.clamp(-1, 1)                                  # deterministic play (players.py:372-388)
                    if "_thrust_x" in name:                                # T = 0.5 + 0.5 a  ->  k T
                        k = float(name.split("_thrust_x")[1])
                        act[:, 3] = (k * (1.0 + act[:, 3]) - 1.0).clamp(-1, 1)
                elif name == "random":
                    act = torch.randn(args.envs, 4, device="cuda", generator=g).clamp(-1, 1)
                else:
                    act = torch.zeros(args.envs, 4, device="cuda")
            obs, rew, dones, infos = env.step(act)
            ep_len += 1
            ep_rew += rew
            rew_sum += float(rew.mean())
            thrust_sum += float(act[:, 3].mean()); speed_sum += float(obs["observation"][:, 6].mean())
            lat_sum += float(obs["observation"][:, 7].mean()); up_sum += float(obs["observation"][:, 8].mean()); act_sum += act.mean(0)
            d = dones.bool()
            if d.any():
                done_len += float(ep_len[d].sum()); done_rew += float(ep_rew[d].sum()); n_done += int(d.sum())
                info = infos["item_reward_info"]
                goal = info["reach_goal_reward"][d] > 0                                        # planning.py:367-368, 380
                # z_reward = min(z - 1.8, 0, 1.2 - z) = -(0.3 + |z - 1.5|) inside the band: outside iff < -0.6 (planning.py:262)
                zout = info["z_reward"][d] < -0.6
                head = info["heading_reward"][d] < 0.25
                n_goal += int(goal.sum())
                causes["altitude_band"] += int((zout & ~goal).sum())
                causes["heading"] += int((head & ~zout & ~goal).sum())
                causes["collision_or_bounds"] += int((~head & ~zout & ~goal).sum())
                ep_len[d] = 0; ep_rew[d] = 0
        out[name] = {"episodes": n_done, "mean_episode_length": round(done_len / max(n_done, 1), 1),
                     "mean_episode_reward": round(done_rew / max(n_done, 1), 2), "mean_reward_per_step": round(rew_sum / args.steps, 4),
                     "goal_reached_fraction": round(n_goal / max(n_done, 1), 4),
                     "termination_causes": {k: round(v / max(n_done, 1), 3) for k, v in causes.items()},
                     "mean_thrust_action": round(thrust_sum / args.steps, 3), "mean_forward_speed": round(speed_sum / args.steps, 3),
                     "mean_lateral_speed": round(lat_sum / args.steps, 3), "mean_vertical_speed": round(up_sum / args.steps, 3),
                     "mean_action": [round(float(v) / args.steps, 3) for v in act_sum]}
        env.env.hip.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
