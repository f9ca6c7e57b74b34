#!/usr/bin/env python3
"""The three launches that are 90 % of a PPO minibatch step, timed IN SEQUENCE (forward -> weight gradient -> dX, as
FusedMLPStep.step issues them, so that each kernel finds the caches as the step leaves them), HIP events around every launch.

    python tools/update_seq_probe.py [--M 196608] [--iters 30] [--recompute 0|1|both]

recompute 0: h1 stored by the forward, read by ag_split_wgrad and ag_split_gemm_input_wgrad (round 4)
recompute 1: h1 never stored; ag_split_wgrad_input and ag_split_gemm_input_wgrad_recompute (round 5)
One JSON line per mode: microseconds per launch (median over iterations) and the algorithmic HBM bytes each launch has to move.
"""
import argparse
import ctypes
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from airgym_amd import _native as N  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=196608)
    ap.add_argument("--D", type=int, default=18)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--recompute", default="both")
    ap.add_argument("--tile-rows", type=int, default=0, help="0 / 256 or 128: row tile of the forward and dX launches (recompute 1 only)")
    a = ap.parse_args()
    lib = N.load()
    M, D, A = a.M, a.D, 4
    g = torch.Generator(device="cuda").manual_seed(0)
    f = dict(device="cuda", dtype=torch.float32)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    obs = torch.randn(M, D, generator=g, **f)
    mean = torch.zeros(D, device="cuda", dtype=torch.float64)
    var = torch.ones(D, device="cuda", dtype=torch.float64)
    W1 = torch.randn(256, D, generator=g, **f) / D ** 0.5
    b1 = 0.1 * torch.randn(256, generator=g, **f)
    W2 = torch.randn(256, 256, generator=g, **f) / 16.0
    b2 = 0.1 * torch.randn(256, generator=g, **f)
    Wh = torch.randn(A + 1, 256, generator=g, **f) / 16.0
    bh = torch.zeros(A + 1, **f)
    image = torch.empty(lib.ag_split_gemm_input_image_bytes(), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_split_gemm_input_prepare(W1.data_ptr(), b1.data_ptr(), D, W2.data_ptr(), image.data_ptr(), st), "in_prepare")
    bwd = torch.empty(lib.ag_split_gemm_plane_bytes(), dtype=torch.uint8, device="cuda")
    N.check(lib.ag_split_gemm_prepare(W2.data_ptr(), bwd.data_ptr(), 256, 256, 1, st), "prepare")
    tiles = M // 128      # (room for 128-row tiles)
    z = lambda *s: torch.zeros(*s, **f)
    keep = {"act": z(M, A), "nlp": z(M), "adv": torch.randn(M, generator=g, **f), "ret": z(M), "val": z(M), "mu": z(M, A),
            "sig": torch.ones(M, A, **f), "lp": z(tiles, lib.ag_ppo_loss_num_sums()), "dwh": z(tiles, A + 1, 256), "db": z(tiles, 256),
            "logstd": z(A)}
    L = N.AgLossEpilogue()
    L.struct_size = ctypes.sizeof(N.AgLossEpilogue)
    L.logstd_dev = keep["logstd"].data_ptr()
    L.actions_dev, L.old_neglogp_dev, L.advantages_dev = keep["act"].data_ptr(), keep["nlp"].data_ptr(), keep["adv"].data_ptr()
    L.returns_dev, L.old_values_dev = keep["ret"].data_ptr(), keep["val"].data_ptr()
    L.old_mu_dev, L.old_sigma_dev, L.new_mu_dev, L.new_sigma_dev = keep["mu"].data_ptr(), keep["sig"].data_ptr(), None, None
    L.heads_dev = None
    L.loss_partials_dev, L.dwh_partials_dev, L.db_partials_dev = keep["lp"].data_ptr(), keep["dwh"].data_ptr(), keep["db"].data_ptr()
    L.e_clip, L.critic_coef, L.bounds_loss_coef, L.clip_value, L.bound_type = 0.2, 2.0, 1e-4, 0, 1
    L.tile_rows = a.tile_rows
    xn, h1, dz = z(M, D), z(M, 256), z(M, 256)
    dw1, db1 = z(tiles, 256, D), z(tiles, 256)
    modes = [0, 1] if a.recompute == "both" else [int(a.recompute)]
    for rc in modes:
        S = lib.ag_split_wgrad_input_slices(M) if rc else lib.ag_split_wgrad_slices(M)
        parts = z(S, 256, 256)
        inp = N.AgInputLayerArgs()
        inp.struct_size, inp.D = ctypes.sizeof(N.AgInputLayerArgs), D
        inp.obs_dev, inp.mean_dev, inp.var_dev, inp.xn_dev = obs.data_ptr(), mean.data_ptr(), var.data_ptr(), xn.data_ptr()
        inp.h1_dev = None if rc else h1.data_ptr()
        inp.eps, inp.clip = 1e-5, 5.0

        def k1():
            N.check(lib.ag_split_gemm_input_loss_heads_bwd(ctypes.byref(inp), image.data_ptr(), b2.data_ptr(), Wh.data_ptr(),
                                                           bh.data_ptr(), dz.data_ptr(), ctypes.byref(L), M, 256, 256, A + 1, st), "k1")

        def k2():
            if rc:
                N.check(lib.ag_split_wgrad_input(dz.data_ptr(), xn.data_ptr(), image.data_ptr(), parts.data_ptr(), M, 256, 256, D, S, st), "k2")
            else:
                N.check(lib.ag_split_wgrad(dz.data_ptr(), h1.data_ptr(), parts.data_ptr(), M, 256, 256, S, st), "k2")

        def k3():
            if rc:
                N.check(lib.ag_split_gemm_input_wgrad_recompute(dz.data_ptr(), bwd.data_ptr(), image.data_ptr(), xn.data_ptr(),
                                                                dw1.data_ptr(), db1.data_ptr(), M, 256, 256, D, a.tile_rows, st), "k3")
            else:
                N.check(lib.ag_split_gemm_input_wgrad(dz.data_ptr(), bwd.data_ptr(), h1.data_ptr(), xn.data_ptr(), dw1.data_ptr(),
                                                      db1.data_ptr(), M, 256, 256, D, st), "k3")
        ks = [k1, k2, k3]
        for _ in range(3):
            for k in ks:
                k()
        s = torch.cuda.current_stream()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(a.iters)]
        for it in range(a.iters):
            ev[it][0].record(s)
            for j, k in enumerate(ks):
                k()
                ev[it][j + 1].record(s)
        torch.cuda.synchronize()
        us = [statistics.median(ev[it][j].elapsed_time(ev[it][j + 1]) * 1e3 for it in range(a.iters)) for j in range(3)]
        mb = 1e-6
        act = 4.0 * M * 256
        io = 4.0 * M * D
        nbytes = ([2 * io + act + (0 if rc else act), act + (io if rc else act) + 4.0 * S * 65536, act + io + (0 if rc else act)])
        names = (["ag_split_gemm_input_loss_heads_bwd (h1 not stored)", "ag_split_wgrad_input", "ag_split_gemm_input_wgrad_recompute"] if rc
                 else ["ag_split_gemm_input_loss_heads_bwd", "ag_split_wgrad", "ag_split_gemm_input_wgrad"])
        print(json.dumps({"recompute_h1": bool(rc), "M": M, "D": D, "iters": a.iters, "tile_rows": a.tile_rows or 256,
                          "kernels": [{"entry_point": n, "us": round(u, 1), "algorithmic_MB": round(b * mb, 1),
                                       "GBps_algorithmic": round(b / u / 1e3, 1)} for n, u, b in zip(names, us, nbytes)],
                          "sum_us": round(sum(us), 1), "sum_algorithmic_MB": round(sum(nbytes) * mb, 1)}), flush=True)


if __name__ == "__main__":
    main()
