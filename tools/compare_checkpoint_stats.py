#!/usr/bin/env python3
"""Statistical cross-check of the (unpinned) depth renderer and state observations against the reference's own simulator:
the input normaliser stored in trained/planning_cnn_rate.pth holds the per-pixel mean/variance of the post-processed depth
image and the mean/variance of the 16-dim state observation AS THE REFERENCE ENV PRODUCED THEM during training.  Compare
with the same statistics of this env (reference policy driving it)."""
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
    ap.add_argument("--steps", type=int, default=600)
    args = ap.parse_args()
    from tools.play_reference_policy import main as _unused  # noqa: F401
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
    ref_img_mean = sd["running_mean_std.running_mean_std.image.running_mean"][0].double()       # [212, 120]
    ref_img_var = sd["running_mean_std.running_mean_std.image.running_var"][0].double()
    ref_obs_mean = sd["running_mean_std.running_mean_std.observation.running_mean"][:16].double()
    ref_obs_var = sd["running_mean_std.running_mean_std.observation.running_var"][:16].double()
    env = vecenv.create_vec_env("planning", args.envs, use_image=True, num_envs=args.envs, ctl_mode="rate", seed=0,
                                sim_device="cuda:0", headless=True)
    obs = env.reset()
    s1 = torch.zeros(212, 120, dtype=torch.float64, device="cuda"); s2 = torch.zeros_like(s1)
    o1 = torch.zeros(16, dtype=torch.float64, device="cuda"); o2 = torch.zeros_like(o1)
    n = 0
    for t in range(args.steps):
        with torch.no_grad():
            mu, _, _ = model.trunk({"image": obs["image"], "observation": obs["observation"]})
            obs, _, _, _ = env.step(mu.clamp(-1, 1))
        img = obs["image"][:, 0].double()
        s1 += img.sum(0); s2 += (img * img).sum(0)
        ob = obs["observation"].double()
        o1 += ob.sum(0); o2 += (ob * ob).sum(0)
        n += args.envs
    my_img_mean = (s1 / n).cpu(); my_img_var = (s2 / n).cpu() - my_img_mean ** 2
    my_obs_mean = (o1 / n).cpu(); my_obs_var = (o2 / n).cpu() - my_obs_mean ** 2

    def corr(a, b):
        a, b = a.flatten() - a.mean(), b.flatten() - b.mean()
        return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    rows = [0, 20, 40, 60, 80, 100, 119]
    out = {
        "image_mean_overall": {"reference": round(float(ref_img_mean.mean()), 4), "this_env": round(float(my_img_mean.mean()), 4)},
        "image_std_overall": {"reference": round(float(ref_img_var.mean().sqrt()), 4), "this_env": round(float(my_img_var.mean().sqrt()), 4)},
        "image_mean_pixelwise_correlation": round(corr(ref_img_mean, my_img_mean), 4),
        "image_mean_by_row_v(top->bottom)": {str(v): [round(float(ref_img_mean[:, v].mean()), 3), round(float(my_img_mean[:, v].mean()), 3)] for v in rows},
        "image_mean_by_column_u(left->right)": {str(u): [round(float(ref_img_mean[u].mean()), 3), round(float(my_img_mean[u].mean()), 3)] for u in (0, 53, 106, 159, 211)},
        "obs_mean[reference, this_env]": [[round(float(a), 3), round(float(b), 3)] for a, b in zip(ref_obs_mean, my_obs_mean)],
        "obs_std[reference, this_env]": [[round(float(a.sqrt()), 3), round(float(b.clamp_min(0).sqrt()), 3)] for a, b in zip(ref_obs_var, my_obs_var)],
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
