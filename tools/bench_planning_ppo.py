#!/usr/bin/env python3
"""End-to-end Planning PPO epoch (BASELINE config 4 shape: depth-image obs + CNN policy, CTBR) on one GPU.
Side measurement for DESIGN.md; the headline bench stays bench.py."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# MIOpen's default find mode benchmarks every solver (including the naive reference convolutions: ~100 s of kernel time at
# 4096-image minibatches, profiles/r01_planning_ppo_kernel_trace.md) the first time a shape is seen; FAST = immediate mode
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch  # noqa: E402
import yaml  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--horizon", type=int, default=24)
    ap.add_argument("--minibatches", type=int, default=24)
    ap.add_argument("--mini-epochs", type=int, default=5)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--encoder", default="cnn", choices=["cnn", "vae"])
    ap.add_argument("--miopen-find", type=int, default=0, help="torch.backends.cudnn.benchmark (MIOpen find mode)")
    ap.add_argument("--encode-chunk", type=int, default=0, help="vae: images per encoder call (0 = all at once)")
    ap.add_argument("--fused-relu-bn", type=int, default=1, help="ReLU + BatchNorm2d pairs on csrc/cnn_kernels.hip")
    ap.add_argument("--graph-update", type=int, default=0, help="minibatch steps as hipGraphs (use_hip_graph_update; opt-in, measured slower)")
    ap.add_argument("--split-fwd", type=int, default=1, help="forward of the 3x3 layers on the bf16 matrix cores (fused_cnn.SPLIT_FWD)")
    args = ap.parse_args()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    params = yaml.safe_load(open(os.path.join(repo, "scripts", "config", "ppo_planning.yaml")))["params"]
    c = params["config"]
    c.update(num_actors=args.envs, horizon_length=args.horizon, mini_epochs=args.mini_epochs,
             minibatch_size=args.envs * args.horizon // args.minibatches, device="cuda:0", max_epochs=-1,
             write_summaries=False, print_stats=False, save_frequency=0, save_best_after=10 ** 9)
    c["env_config"] = {"use_image": True, "num_envs": args.envs, "ctl_mode": "rate", "seed": 0, "sim_device": "cuda:0",
                       "headless": True}
    if args.encoder == "vae":
        params["network"].pop("cnn", None)
        params["network"]["vae"] = {"latent_dims": 64, "allow_random_init": True, "image_res": [120, 212],
                                    "interpolation_mode": "bilinear", "return_sampled_latent": False,
                                    "encode_chunk": args.encode_chunk}
    params["seed"] = 0
    c["use_hip_graph_update"] = bool(args.graph_update)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    from airgym_amd.lib.agent.a2c_continuous import A2CAgent
    from airgym_amd.lib.network import fused_cnn
    fused_cnn.SPLIT_FWD = bool(args.split_fwd)
    agent = A2CAgent("planning_bench", params)
    # (channels_last was tried: MIOpen falls back to its naive kernels for these shapes, 137 s per epoch instead of 5.7 s)
    for mod in agent.model.modules():
        if hasattr(mod, "fused_relu_bn"):
            mod.fused_relu_bn = bool(args.fused_relu_bn)
    agent.init_tensors()
    agent.obs = agent.env_reset()
    for _ in range(args.warmup):
        agent.epoch_num += 1
        agent.train_epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    play = upd = 0.0
    for _ in range(args.steps):
        agent.epoch_num += 1
        st = agent.train_epoch()
        play += st["play_time"]; upd += st["update_time"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": f"env_steps_per_sec_planning_{args.envs}_envs_per_gpu", "encoder": args.encoder,
                      "value": args.envs * args.horizon * args.steps / dt, "ms_per_epoch": dt / args.steps * 1e3,
                      "rollout_ms": play / args.steps * 1e3, "update_ms": upd / args.steps * 1e3,
                      "minibatch": c["minibatch_size"], "mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
                      "kl": st["kl"], "a_loss": st["a_loss"], "c_loss": st["c_loss"], "graph_update": bool(getattr(agent, "_graph_generic", False)), "split_fwd": bool(args.split_fwd),
                      "graphs": len(getattr(agent, "_upd_graphs", {})), "graph_error": getattr(agent, "_graph_generic_error", None)}))


if __name__ == "__main__":
    main()
