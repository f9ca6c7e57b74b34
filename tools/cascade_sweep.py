#!/usr/bin/env python3
"""Structural identification of the (unpinned) rate cascade against the reference's own trained Planning policy
(VERDICT r02 next #5): fly trained/planning_cnn_rate.pth in THIS env under axis-sign hypotheses for the body-rate command
(FLU vs FRD: the policy's roll / pitch / yaw rate set-points negated before they reach the env, and un-negated in the action
echo of the observation, obs[12:15]), and score each by episode length, goal-reach fraction, forward / lateral speed and the
distance of the 16-dim observation statistics to the input normaliser stored in the checkpoint (= what the reference simulator
produced during training).  The world- vs body-frame hypothesis for the angular velocity handed to the rate loop is a compile-time
variant of the experiments build (-DAG_EXP_CASCADE=1: tools/cascade_sweep.py runs under whatever library AIRGYM_EXP_LIB names).

    python tools/cascade_sweep.py --checkpoint /root/reference/trained/planning_cnn_rate.pth   (this container only; the file is never copied) --signs +++ -++ +-+ ++- --+ -+- +-- ---
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--envs", type=int, default=512)
    ap.add_argument("--steps", type=int, default=1200)
    ap.add_argument("--signs", nargs="+", default=["+++", "-++", "+-+", "++-", "--+", "-+-", "+--", "---"])
    ap.add_argument("--tag", default="default")
    args = ap.parse_args()
    from airgym_amd.lib.model.a2c_continuous_logstd_model import ModelA2CContinuousLogStd
    from airgym_amd.lib.utils import vecenv
    import airgym_amd.envs  # noqa: F401
    params = {"network": {"separate": False, "mlp": {"units": [64, 128, 64], "activation": "elu"},
                          "space": {"continuous": {"fixed_sigma": True}}, "cnn": {"output_dim": 30}},
              "config": {"normalize_input": True, "normalize_value": True}}
    keys = {"actions_num": 4, "input_shape": {"image": (1, 212, 120), "observation": (16,)}}
    model = ModelA2CContinuousLogStd(params, keys)
    sd = torch.load(args.checkpoint, map_location="cpu", weights_only=False)["model"]
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    ref_mean = sd["running_mean_std.running_mean_std.observation.running_mean"][:16].double().cuda()
    ref_std = sd["running_mean_std.running_mean_std.observation.running_var"][:16].double().sqrt().cuda()
    for signs in args.signs:
        sg = torch.tensor([1.0 if c in "+p" else -1.0 for c in signs] + [1.0], device="cuda")      # "+-+" or "pnp"
        env = vecenv.create_vec_env("planning", args.envs, use_image=True, num_envs=args.envs, ctl_mode="rate", seed=0,
                                    sim_device="cuda:0", headless=True)
        obs = env.reset()
        ep_len = torch.zeros(args.envs, device="cuda"); ep_rew = torch.zeros(args.envs, device="cuda")
        done_len = done_rew = 0.0
        n_done = n_goal = 0
        o1 = torch.zeros(16, dtype=torch.float64, device="cuda"); o2 = torch.zeros_like(o1)
        act_sum = torch.zeros(4, device="cuda")
        for t in range(args.steps):
            with torch.no_grad():
                ob = obs["observation"].clone()
                ob[:, 12:16] *= sg                                   # the policy sees its OWN action echoed back
                mu, _, _ = model.trunk({"image": obs["image"], "observation": ob})
                act = mu.clamp(-1, 1)
            obs, rew, dones, infos = env.step(act * sg)
            ep_len += 1; ep_rew += rew; act_sum += act.mean(0)
            o = ob.double(); o1 += o.sum(0); o2 += (o * o).sum(0)
            d = dones.bool()
            if d.any():
                done_len += float(ep_len[d].sum()); done_rew += float(ep_rew[d].sum()); n_done += int(d.sum())
                n_goal += int((infos["item_reward_info"]["reach_goal_reward"][d] > 0).sum())
                ep_len[d] = 0; ep_rew[d] = 0
        n = args.steps * args.envs
        mean = o1 / n
        std = (o2 / n - mean * mean).clamp_min(0).sqrt()
        z = ((mean - ref_mean).abs() / ref_std.clamp_min(1e-6))
        out = {"library": args.tag, "rate_signs(roll,pitch,yaw)": signs, "episodes": n_done,
               "mean_episode_length": round(done_len / max(n_done, 1), 1), "mean_episode_reward": round(done_rew / max(n_done, 1), 1),
               "goal_reached_fraction": round(n_goal / max(n_done, 1), 4),
               "forward_speed": round(float(mean[6]), 3), "lateral_speed": round(float(mean[7]), 3), "vertical_speed": round(float(mean[8]), 3),
               "reference_forward_speed": round(float(ref_mean[6]), 3), "reference_lateral_speed": round(float(ref_mean[7]), 3),
               "mean_policy_action": [round(float(v) / args.steps, 3) for v in act_sum],
               "obs_mean_distance_sum_z": round(float(z.sum()), 3), "obs_mean_z_by_dim": [round(float(v), 2) for v in z],
               "obs_std_ratio": [round(float(a / b.clamp_min(1e-6)), 2) for a, b in zip(std, ref_std)]}
        print(json.dumps(out), flush=True)
        env.env.hip.close()


if __name__ == "__main__":
    main()
