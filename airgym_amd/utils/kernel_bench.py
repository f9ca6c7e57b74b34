"""Measure the env-step kernel in isolation: HIP events around a replayed hipGraph of K launches.

Used by bench.py (the `roofline` object) and by tools/sweep_env_kernel.py.  Events are recorded on the
stream the kernels are launched on (torch's current stream is handed to ag_step as the hipStream_t).
"""
import torch

# algorithmic HBM bytes per env-step, fp32 SoA, state + action + controller
# memory in, state + obs + reward + flags out.
ALGO_BYTES_PER_ENV_STEP = {
    ("hovering", "rate"): 287,
    ("tracking", "vel"): 543,
}
HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def algo_bytes(task, ctl_mode, num_obs, num_actions):
    key = (task, ctl_mode)
    if key in ALGO_BYTES_PER_ENV_STEP:
        return ALGO_BYTES_PER_ENV_STEP[key]
    ctl_floats = {"prop": 0, "rate": 6, "atti": 6, "vel": 12, "pos": 12}[ctl_mode]
    reads = 13 * 4 + num_actions * 4 * 2 + ctl_floats * 4 + 4 + 1
    writes = 13 * 4 + num_actions * 4 + ctl_floats * 4 + 4 + num_obs * 4 + 4 + 1 + 1
    return reads + writes


KERNEL_VARIANTS = {0: "step_kernel_ws2<{t},{c},false,false>", 1: "step_kernel_ws<{t},{c}>", 2: "step_kernel_ws2<{t},{c},true,false>",
                   3: "step_kernel_ws2<{t},{c},true,true>", 4: "step_kernel_ws2<{t},{c},false,true>"}
_TASK_ID = {"hovering": 0, "tracking": 1}
_CTL_ID = {"pos": 0, "vel": 1, "atti": 2, "rate": 3, "prop": 4}


def kernel_name(task, ctl_mode, variant=0):
    """Symbol (as rocprofv3 prints it) of the env-step kernel a handle launches for launch-params variant `variant`."""
    fmt = KERNEL_VARIANTS.get(variant, "step_kernel<{t},{c}," + str(variant) + ",...>")
    return "ag::" + fmt.format(t=_TASK_ID[task], c=_CTL_ID[ctl_mode])


def measure_env_kernel(env, steps_per_graph=48, replays=52, warmup_replays=3, use_graph=True, seed=1, rollout_form=True):
    """Returns dict(us_per_step, env_steps_per_s, gbps_algorithmic, ...).  `env` is a HipEnvHandle.

    rollout_form=True times the kernel exactly as the PPO rollout launches it (ag_step_rollout: obs / reward / u8 done
    flags into rollout slots, per-tile reward-term sums); False times the drop-in ag_step (int64 reset_buf, per-env
    item_reward_info arrays).  Default 48 x 52 = 2 496 steps, so the 2 400-step time limit fires inside the timed
    region."""
    assert steps_per_graph % 2 == 0, "capture an even number of steps (device tick ping-pong)"
    dev = env.device
    n, A = env.num_envs, env.num_actions
    g = torch.Generator(device=dev).manual_seed(seed)
    # what a freshly initialised policy emits: N(0,1) clamped to [-1,1]
    actions = torch.randn(steps_per_graph, n, A, generator=g, device=dev).clamp_(-1.0, 1.0)
    H = 24
    if rollout_form:
        obs = torch.zeros(H + 1, n, env.num_obs, device=dev)
        rew = torch.zeros(H, n, device=dev)
        done = torch.zeros(H + 1, n, dtype=torch.uint8, device=dev)
        tiles = torch.zeros(H, (n + 63) // 64, 12, device=dev)

    def one(t):
        if rollout_form:
            s = t % H
            env.step_rollout(actions[t % steps_per_graph], obs[s + 1], rew[s], done[s + 1], tiles[s])
        else:
            env.step(actions[t % steps_per_graph])
    stream = torch.cuda.Stream(device=dev)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        for t in range(4):
            one(t)
        stream.synchronize()
        graph = None
        if use_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
                for t in range(steps_per_graph):
                    one(t)

        def run():
            if graph is not None:
                graph.replay()
            else:
                for t in range(steps_per_graph):
                    one(t)
        for _ in range(warmup_replays):
            run()
        stream.synchronize()
        start.record(stream)
        for _ in range(replays):
            run()
        stop.record(stream)
        stop.synchronize()
    ms = start.elapsed_time(stop)
    total_steps = steps_per_graph * replays
    us = ms * 1e3 / total_steps
    b = algo_bytes(env.task, env.ctl_mode, env.num_obs, A)
    return {
        "us_per_step": us,
        "env_steps_per_s": n / (us * 1e-6),
        "gbps_algorithmic": n * b / (us * 1e-6) / 1e9,
        "algo_bytes_per_env_step": b,
        "graph": bool(use_graph),
        "steps_timed": total_steps,
        "form": "ag_step_rollout" if rollout_form else "ag_step",
    }


def measure_copy_ceiling(device, nbytes=1 << 30, iters=20):
    """Achievable HBM bandwidth of this device: a device-to-device copy of `nbytes` (read + write counted), GB/s."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    src.fill_(1)
    us = _time_us(lambda: dst.copy_(src), iters=iters, warmup=3)
    return 2.0 * nbytes / us / 1e3


BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (never the 2:1-sparsity figure)
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA = f32 vector rate, 64 FLOP/clk/SIMD


def _time_us(fn, iters=20, warmup=3):
    s = torch.cuda.current_stream()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(warmup):
        fn()
    start.record(s)
    for _ in range(iters):
        fn()
    stop.record(s)
    stop.synchronize()
    return start.elapsed_time(stop) * 1e3 / iters


@torch.no_grad()
def measure_update_kernels(agent, iters=20):
    """Per-kernel roofline figures of the PPO minibatch (hand-scheduled path): the hipBLASLt fp32 GEMM against the f32-MFMA
    peak and the HIP streaming kernels against HBM, timed with events on the launch stream, on the live activations."""
    import ctypes

    from airgym_amd import _native as N
    fs = getattr(agent, "_fused_step", None)
    if fs is None or len(fs.layers) < 2:
        return []
    lib = N.load()
    st = ctypes.c_void_p(torch.cuda.current_stream(agent.ppo_device).cuda_stream)
    M, A1 = fs.M, fs.A + 1
    w, b = fs.layers[-1][0], fs.layers[-1][1]
    C, K = w.shape
    x, h = fs.h[-2], fs.h[-1]
    out = []
    if fs.fuse_heads:       # what the step runs: NN product against the transposed weight copy (rocBLAS / hipBLASLt via TunableOp)
        fs.wt_last.copy_(w.t())
        us = _time_us(lambda: torch.mm(x, fs.wt_last, out=fs.dz[:M * C].view(M, C)), iters)
        label = f"library f32 GEMM (NN) [{M}x{K}]x[{K}x{C}] (the forward product under --split-gemm 0; for reference)"
    else:
        us = _time_us(lambda: torch.addmm(b, x, w.t(), out=fs.dz[:M * C].view(M, C)), iters)
        label = f"library f32 GEMM (TN + bias) [{M}x{K}]x[{K}x{C}] (the forward product under --split-gemm 0; for reference)"
    flops = 2.0 * M * C * K
    out.append({"kernel": label, "bound": "mfma",
                "achieved": flops / us / 1e6, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": flops / us / 1e6 / FP32_MFMA_PEAK_TFLOPS, "us_per_launch": us,
                "note": "timed as 20 back-to-back launches (sustained-MFMA clocks); inside the minibatch, between HBM-bound "
                        "kernels, the same GEMM takes 185-205 us = 126-140 TFLOP/s (profiles/r01_bench_fused_kernel_trace.md)"})
    # the weight gradient of the 256 x 256 layer as the step runs it: split-K batched GEMM of the library (f32 MFMA)
    S = fs.wgrad_partials[-1].shape[0]
    if M % S == 0:
        dzv, xv = fs.dz[:M * C].view(S, M // S, C).transpose(1, 2), x.view(S, M // S, K)
        us = _time_us(lambda: torch.bmm(dzv, xv, out=fs.wgrad_partials[-1]), iters)
        out.append({"kernel": f"library f32 split-K weight gradient [{C}x{M}]x[{M}x{K}] ({S} slices)", "bound": "mfma",
                    "achieved": flops / us / 1e6, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": flops / us / 1e6 / FP32_MFMA_PEAK_TFLOPS, "us_per_launch": us,
                    "note": "the one GEMM of the update still at the f32 matrix-core rate (90 % pipe occupancy, "
                            "profiles/r02_update_kernels_pmc.md)"})
    if getattr(fs, "split", None):
        sg = next(iter(fs.split.values()))
        sg.prepare()

        def mfma(name, us, fl, note):
            out.append({"kernel": name, "bound": "mfma", "achieved": 6.0 * fl / us / 1e6, "peak": BF16_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": 6.0 * fl / us / 1e6 / BF16_MFMA_PEAK_TFLOPS, "us_per_launch": us,
                        "f32_equivalent_tflops": fl / us / 1e6, "note": note})
        peak_note = ("peak = dense bf16 MFMA (2.5 PFLOP/s); the f32-equivalent rate is what replaces the f32-MFMA GEMM "
                     "(157.3 TFLOP/s peak)")
        us = _time_us(lambda: sg.forward(x, fs.dz[:M * C].view(M, C)), iters)
        mfma(f"ag_split_gemm [{M}x{K}]x[{K}x{C}] (6 bf16 MFMAs per f32 product, f32-accurate), plain", us, flops, peak_note)
        if getattr(fs, "fuse_gemm_heads", False) and fs.fuse_heads:
            us = _time_us(lambda: sg.forward_elu_heads(x, fs.dz[:M * C].view(M, C), b, agent.heads_w, agent.heads_b, fs.heads), iters)
            mfma("ag_split_gemm_elu_heads (update forward of the last hidden layer + ELU + heads, as the step runs it)", us,
                 flops + 2.0 * M * C * A1, "replaces ag_split_gemm + ag_elu_heads (one pass over z less)")
        if getattr(fs, "fuse_gemm_input_wgrad", False):
            D0 = fs.layers[0][0].shape[1]
            us = _time_us(lambda: sg.backward_input_wgrad(h, fs.h[0], fs.xn, fs.wgrad_partials[0], fs.bias_partials[0]), iters)
            mfma("ag_split_gemm_input_wgrad (dX of layer 2 + ELU' + dW1 / db1 on the matrix cores, as the step runs it)", us,
                 flops + 2.0 * M * C * 32, "replaces ag_split_gemm + ag_elu_bwd_input_wgrad; dh1 / dz1 never written; the "
                 "epilogue's K = rows products are counted at their padded size (32 input columns)")
    scratch = fs.dz[:M * C].view(M, C)
    scratch.copy_(h)

    def hbm(name, us, nbytes):
        out.append({"kernel": name, "bound": "hbm", "achieved": nbytes / us / 1e3, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": nbytes / us / 1e3 / HBM_PEAK_GBPS, "us_per_launch": us, "algo_bytes": nbytes})
    us = _time_us(lambda: lib.ag_elu_heads(scratch.data_ptr(), agent.heads_w.data_ptr(), agent.heads_b.data_ptr(),
                                            fs.heads.data_ptr(), M, C, A1, 0, None, st), iters)
    hbm("ag_elu_heads (ELU + head product, pre-activation kept)" + (" - folded into the GEMM epilogue in the step, shown for reference"
        if getattr(fs, "fuse_gemm_heads", False) and getattr(fs, "split", None) else ""), us, 4.0 * M * (C + A1))
    parts = fs.bias_partials[-1]
    us = _time_us(lambda: lib.ag_heads_bwd_elu_wgrad(fs.d_heads.data_ptr(), agent.heads_w.data_ptr(), h.data_ptr(),
                                                      scratch.data_ptr(), parts.data_ptr(), fs.head_wg_partials.data_ptr(),
                                                      M, C, A1, 1, None, st), iters)
    hbm("ag_heads_bwd_elu_wgrad (head dX + ELU' + head wgrad)", us, 4.0 * M * (2 * C + A1))
    if fs.fuse_input_wgrad and lib.ag_input_wgrad_rows(fs.layers[0][0].shape[1]) > 0:
        D = fs.layers[0][0].shape[1]
        C0 = fs.layers[0][0].shape[0]
        iparts = fs.bias_partials[0]
        us = _time_us(lambda: lib.ag_elu_bwd_input_wgrad(fs.dh.data_ptr(), fs.h[0].data_ptr(), fs.xn.data_ptr(),
                                                          fs.wgrad_partials[0].data_ptr(), iparts.data_ptr(), M, C0, D, st), iters)
        hbm("ag_elu_bwd_input_wgrad (ELU' + first-layer wgrad)" + (" - folded into the dX GEMM's epilogue in the step, shown for reference"
            if getattr(fs, "fuse_gemm_input_wgrad", False) else ""), us, 4.0 * M * (2 * C0 + D))
    return out
