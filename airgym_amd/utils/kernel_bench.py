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


_TASK_ID = {"hovering": 0, "tracking": 1}
_CTL_ID = {"pos": 0, "vel": 1, "atti": 2, "rate": 3, "prop": 4}


def kernel_name(task, ctl_mode, fused=False, single=False):
    """Symbol (as rocprofv3 prints it) of the env-step kernel (csrc/step_kernel.hip): step_kernel_ws2<task, ctl, true> for
    ag_step_rollout_fused, step_kernel_ws2<task, ctl, false> for the one-step launches (ag_step / ag_step_into / ag_step_rollout),
    step_kernel_multi<task, ctl> for ag_step_multi."""
    if fused:
        return "ag::step_kernel_ws2<%d,%d,true>" % (_TASK_ID[task], _CTL_ID[ctl_mode])
    if single:
        return "ag::step_kernel_ws2<%d,%d,false>" % (_TASK_ID[task], _CTL_ID[ctl_mode])
    return "ag::step_kernel_multi<%d,%d>" % (_TASK_ID[task], _CTL_ID[ctl_mode])


def fused_algo_bytes(task, ctl_mode, num_obs, num_actions):
    """Algorithmic bytes per env-step of ag_step_rollout_fused = the env step's minus the action read (the action
    goes from the sampler to the integrator through LDS) plus what the rollout head and tail move: heads in (A+1 f32),
    actions / mus / sigmas (A f32 each), neglogp, values, shaped reward out, running episode reward / shaped reward / length
    read and written."""
    A = num_actions
    return algo_bytes(task, ctl_mode, num_obs, A) - 4 * A + 4 * (A + 1) + 3 * 4 * A + 3 * 4 + 2 * 3 * 4


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


def multi_own_bytes(task, ctl_mode, num_obs, num_actions, K):
    """Bytes per env-step ag_step_multi really has to move at K steps per launch (SURVEY 8(d)'s accounting, un-padded): per
    step the action in, observation / reward / u8 done out; the state + controller memory + previous action + progress + flag
    in and out ONCE per launch."""
    ctl_floats = {"prop": 0, "rate": 6, "atti": 6, "vel": 12, "pos": 12}[ctl_mode]
    state = 13 * 4 + num_actions * 4 + ctl_floats * 4 + 4 + 1
    return num_actions * 4 + num_obs * 4 + 4 + 1 + (2 * state + 1) / float(K)


def measure_env_multi(env, K=24, launches_per_graph=2, replays=52, warmup_replays=3, seed=1, use_graph=True):
    """ag_step_multi: K env steps per launch (state in registers between them), `launches_per_graph` launches per hipGraph
    (even: device tick ping-pong), HIP events on the launch stream.  Default 24 x 2 x 52 = 2 496 env steps, the same count
    (and the same N(0,1)-clamped actions) as measure_env_kernel, so the 2 400-step time limit fires inside the timed region."""
    assert launches_per_graph % 2 == 0
    dev = env.device
    n, A = env.num_envs, env.num_actions
    g = torch.Generator(device=dev).manual_seed(seed)
    actions = torch.randn(launches_per_graph, K, n, A, generator=g, device=dev).clamp_(-1.0, 1.0)
    obs = torch.zeros(K, n, env.num_obs, device=dev)
    rew = torch.zeros(K, n, device=dev)
    done = torch.zeros(K, n, dtype=torch.uint8, device=dev)
    tiles = torch.zeros(K, (n + 63) // 64, 12, device=dev)

    def one(j):
        env.step_multi(actions[j % launches_per_graph], obs, rew, done, None, tiles)
    stream = torch.cuda.Stream(device=dev)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        for j in range(2):
            one(j)
        stream.synchronize()
        graph = None
        if use_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
                for j in range(launches_per_graph):
                    one(j)

        def run():
            if graph is not None:
                graph.replay()
            else:
                for j in range(launches_per_graph):
                    one(j)
        for _ in range(warmup_replays):
            run()
        stream.synchronize()
        start.record(stream)
        for _ in range(replays):
            run()
        stop.record(stream)
        stop.synchronize()
    launches = launches_per_graph * replays
    us_launch = start.elapsed_time(stop) * 1e3 / launches
    us_step = us_launch / K
    b = algo_bytes(env.task, env.ctl_mode, env.num_obs, A)
    own = multi_own_bytes(env.task, env.ctl_mode, env.num_obs, A, K)
    return {"us_per_launch": us_launch, "us_per_step": us_step, "steps_per_launch": K, "launches_timed": launches,
            "env_steps_timed": launches * K, "env_steps_per_s": n / (us_step * 1e-6),
            "gbps_algorithmic": n * b / (us_step * 1e-6) / 1e9, "algo_bytes_per_env_step": b,
            "gbps_own_bytes": n * own / (us_step * 1e-6) / 1e9, "own_bytes_per_env_step": own, "form": "ag_step_multi"}


@torch.no_grad()
def measure_fused_rollout_kernel(agent, steps_per_graph=48, replays=52, warmup_replays=3):
    """ag_step_rollout_fused exactly as FusedRolloutStep launches it (policy sampling + env step + accounting, heads held
    fixed at what the freshly initialised policy emits), timed like measure_env_kernel: hipGraph of launches, HIP events on the
    launch stream.  Uses scratch rollout slots so that the agent's own buffers are not disturbed."""
    import ctypes

    from airgym_amd import _native as N
    fr, env = agent._fused_rollout, agent._hip_env
    dev, n, A, H = env.device, env.num_envs, env.num_actions, agent.horizon_length
    f = dict(device=dev, dtype=torch.float32)
    obs = torch.zeros(H + 1, n, env.num_obs, **f)
    rew = torch.zeros(H, n, **f)
    done = torch.zeros(H + 1, n, dtype=torch.uint8, device=dev)
    tiles = torch.zeros(H, (n + 63) // 64, 12, **f)
    acts, mus, sig = (torch.zeros(H, n, A, **f) for _ in range(3))
    nlp, val, shp = (torch.zeros(H, n, **f) for _ in range(3))
    cur = torch.zeros(3, n, **f)
    parts = torch.zeros(H, (n + 63) // 64, 4, dtype=torch.float64, device=dev)
    heads = torch.zeros(n, A + 1, **f)               # mu = 0 (freshly initialised policy), value 0
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    tails = []
    for s in range(H):
        src = fr._tail(s)
        t = N.AgRolloutTail()
        ctypes.memmove(ctypes.byref(t), ctypes.byref(src), ctypes.sizeof(t))
        t.heads_dev, t.counter_dev = heads.data_ptr(), counter.data_ptr()
        t.actions_dev, t.mus_dev, t.sigmas_dev = acts[s].data_ptr(), mus[s].data_ptr(), sig[s].data_ptr()
        t.neglogp_dev, t.values_dev, t.shaped_dev = nlp[s].data_ptr(), val[s].data_ptr(), shp[s].data_ptr()
        t.cur_rew_dev, t.cur_shaped_dev, t.cur_len_dev = cur[0].data_ptr(), cur[1].data_ptr(), cur[2].data_ptr()
        t.partials_dev = parts[s].data_ptr()
        tails.append(t)

    def one(k):
        s = k % H
        env.step_rollout_fused(tails[s], obs[s + 1], rew[s], done[s + 1], tiles[s])
    stream = torch.cuda.Stream(device=dev)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        for k in range(4):
            one(k)
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
            for k in range(steps_per_graph):
                one(k)
        for _ in range(warmup_replays):
            graph.replay()
        stream.synchronize()
        start.record(stream)
        for _ in range(replays):
            graph.replay()
        stop.record(stream)
        stop.synchronize()
    total = steps_per_graph * replays
    us = start.elapsed_time(stop) * 1e3 / total
    b = fused_algo_bytes(env.task, env.ctl_mode, env.num_obs, A)
    return {"us_per_step": us, "env_steps_per_s": n / (us * 1e-6), "gbps_algorithmic": n * b / (us * 1e-6) / 1e9,
            "algo_bytes_per_env_step": b, "steps_timed": total, "kernel": kernel_name(env.task, env.ctl_mode, True),
            "form": "ag_step_rollout_fused"}


def measure_copy_ceiling(device, nbytes=1 << 30, iters=20):
    """Achievable HBM bandwidth of this device: a device-to-device copy of `nbytes` (read + write counted), GB/s."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    src.fill_(1)
    us = _time_us(lambda: dst.copy_(src), iters=iters, warmup=3)
    return 2.0 * nbytes / us / 1e3


def env_kernel_source_sha():
    """Provenance key of the env-step kernel: sha256 over the sources it is compiled from.  The PMC traffic figure is read
    from a committed file (counters cannot be read live); it is only quoted when the file was measured on THESE sources."""
    import hashlib
    import os
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
    h = hashlib.sha256()
    for f in ("step_kernel.hip", "env_math.hpp", "kernel_args.hpp", "rollout_math.hpp"):
        with open(os.path.join(here, f), "rb") as fh:
            h.update(fh.read())
    # ... and the flags those sources are compiled with (build.py's own text would tie the key to every unrelated unit it lists)
    import importlib.util
    spec = importlib.util.spec_from_file_location("_airgym_build", os.path.join(here, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    flags = sorted({" ".join(d) for (_, src, d) in b.units() if src == "step_kernel.hip" and "-DAG_TASK=0" in d and "-DAG_CTL=3" in d})
    h.update((" ".join(b.COMMON[1:]) + " | " + " ; ".join(flags)).encode())
    return h.hexdigest()[:16]


def pmc_traffic(repo, key, kernel):
    """(bytes per launch, source note) from profiles/*_env_kernel_pmc.json, newest round first - or (None, why not)."""
    import glob
    import json
    import os
    sha = env_kernel_source_sha()
    why = "no PMC record for this kernel"
    for path in sorted(glob.glob(os.path.join(repo, "profiles", "r*_env_kernel_pmc.json")), reverse=True):
        rec = json.load(open(path)).get(key)
        if not rec or rec.get("kernel") != kernel:
            continue
        if rec.get("source_sha") != sha:
            why = (f"stale: {os.path.basename(path)} was measured on kernel sources {rec.get('source_sha')}, this build is {sha} "
                   f"(re-run tools/gpu_pmc_env.sh)")
            continue
        return rec["traffic_bytes_per_launch"], rec["source"]
    return None, why


def roofline_object(agent, hip, args, repo):
    """bench.py's `env_kernels` (+ `env_only`): the env-step kernels of this configuration, each timed as a hipGraph of launches
    with HIP events on the launch stream (2 496 env steps, so the 2 400-step time limit fires inside the timed region) and priced
    against HBM.  Three forms:
      in_loop     - what the PPO rollout of the timed region launches (ag_step_rollout_fused, one env step per launch behind the
                    policy forward), at ITS algorithmic bytes (375 B for Hovering / CTBR);
      single_step - ag_step_rollout (one step per launch, no policy sample / accounting), SURVEY 8(d)'s 287 B;
      multi_step  - ag_step_multi, K = 24 steps per launch with the state in registers (the env-only metric): priced at the
                    bytes it really moves (101 B per env-step); the launch is vector-ALU bound, not HBM bound
                    (`valu_busy_pct` from the committed SQ counters), `frac_nominal` = the same at 8(d)'s 287 B for continuity.
    `traffic` = HBM bytes per launch from the committed PMC record, quoted only when the record was measured on these sources."""
    task, ctl = args.task, args.ctl
    fr = getattr(agent, "_fused_rollout", None)
    fused = bool(getattr(fr, "fuse_tail", False))
    K = 24
    m = measure_env_multi(hip, K=K, launches_per_graph=2, replays=52)
    r = measure_env_kernel(hip, steps_per_graph=48, replays=52, rollout_form=True)
    r_api = measure_env_kernel(hip, steps_per_graph=48, replays=10, rollout_form=False)
    none = (None, "PMC passes exist for 65 536 envs per launch only")
    traffic, tsrc = none
    traffic1, tsrc1 = none
    valu = None
    if args.envs == 65536:
        traffic, tsrc = pmc_traffic(repo, f"{task}_{ctl}_multi{K}", kernel_name(task, ctl, False))
        traffic1, tsrc1 = pmc_traffic(repo, f"{task}_{ctl}", kernel_name(task, ctl, False, True))
        valu = pmc_field(repo, f"{task}_{ctl}_multi{K}", kernel_name(task, ctl, False), "valu_busy_pct")
    copy_gbps = measure_copy_ceiling(agent.ppo_device)
    n = args.envs
    kernels = {
        "single_step": {
            "bound": "hbm", "kernel": kernel_name(task, ctl, False, True), "entry_point": "ag_step_rollout",
            "us_per_launch": r["us_per_step"], "launches_timed": r["steps_timed"], "algo_bytes_per_env_step": r["algo_bytes_per_env_step"],
            "achieved": r["gbps_algorithmic"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": r["gbps_algorithmic"] / HBM_PEAK_GBPS,
            "traffic": traffic1, "traffic_source": tsrc1,
            "drop_in_ag_step_us": r_api["us_per_step"]},
        "multi_step": {
            "bound": "valu", "kernel": kernel_name(task, ctl, False), "entry_point": f"ag_step_multi ({K} env steps per launch)",
            "us_per_launch": m["us_per_launch"], "launches_timed": m["launches_timed"], "steps_per_launch": K,
            "us_per_env_step_batch": m["us_per_step"], "algo_bytes_per_env_step": m["own_bytes_per_env_step"],
            "achieved": m["gbps_own_bytes"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": m["gbps_own_bytes"] / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": tsrc, "valu_busy_pct": valu,
            "frac_nominal": m["gbps_algorithmic"] / HBM_PEAK_GBPS, "nominal_bytes_per_env_step": m["algo_bytes_per_env_step"],
            "note": "state in registers between the K steps: moves 101 B per env-step, not 8(d)'s 287 B; vector-ALU bound"},
        "copy_ceiling_gbps": copy_gbps, "kernel_source_sha": env_kernel_source_sha(), "envs_per_launch": n,
    }
    out = {"env_kernels": kernels,
           "env_only": {"value": m["env_steps_per_s"], "unit": "env-steps/s",
                        "note": f"env-step kernel only, ag_step_multi ({K} steps per launch), synthetic N(0,1) clamped "
                                f"actions pre-generated on the device, hipGraph replay",
                        "single_step_launch": r["env_steps_per_s"]}}
    if fused:
        rf = measure_fused_rollout_kernel(agent, steps_per_graph=48, replays=52)
        ftraffic, fsrc = none
        if args.envs == 65536:
            ftraffic, fsrc = pmc_traffic(repo, f"{task}_{ctl}_fused", rf["kernel"])
        kernels["in_loop"] = {
            "bound": "hbm", "kernel": rf["kernel"], "entry_point": "ag_step_rollout_fused",
            "us_per_launch": rf["us_per_step"], "launches_timed": rf["steps_timed"],
            "algo_bytes_per_env_step": rf["algo_bytes_per_env_step"],
            "achieved": rf["gbps_algorithmic"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": rf["gbps_algorithmic"] / HBM_PEAK_GBPS,
            "traffic": ftraffic, "traffic_source": fsrc, "frac_of_copy_ceiling": rf["gbps_algorithmic"] / copy_gbps,
            "note": "what the PPO rollout of the timed region launches: policy sample + env step + episode accounting, one env "
                    "step per launch behind the policy forward"}
    else:
        kernels["in_loop"] = dict(kernels["single_step"], note="the rollout of this configuration launches the plain one-step form")
    return out


def update_source_sha():
    """Provenance key of the update's matrix-core kernels (the sources + flags they are compiled from), as env_kernel_source_sha."""
    import hashlib
    import os
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
    h = hashlib.sha256()
    for f in ("split_gemm.hip", "split_wgrad.hip", "split_common.hpp", "ppo_loss_math.hpp"):
        with open(os.path.join(here, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def update_pmc_record(repo, entry_point):
    """(record, source note) of one update launch from profiles/r*_update_kernels_pmc.json, newest round first; the record is
    returned only when it was measured on the kernel sources of this tree (`update_source_sha`)."""
    import glob
    import json
    import os
    sha = update_source_sha()
    why = "no PMC record for this kernel"
    for path in sorted(glob.glob(os.path.join(repo, "profiles", "r*_update_kernels_pmc.json")), reverse=True):
        rec = json.load(open(path)).get(entry_point)
        if not rec:
            continue
        if rec.get("source_sha") != sha:
            why = (f"stale: {os.path.basename(path)} was measured on kernel sources {rec.get('source_sha')}, this build is {sha} "
                   f"(re-run tools/gpu_pmc_update.sh)")
            continue
        return rec, rec.get("source", os.path.basename(path))
    return None, why


def update_roofline(agent, repo, seq=None, optimizer_steps_per_epoch=None):
    """bench.py's top-level `roofline`: the dominant kernel of the timed region (the update's forward + loss launch, the largest
    share of an optimizer step), bound by the bf16 matrix cores.  `achieved` = algorithmic matrix-core FLOPs of ONE launch (every
    f32 product = 6 bf16 MFMAs, incl. the first-layer product the launch carries: 2 M (32 + 256) 256 x 6) / the launch's duration,
    HIP events on the launch stream with the step's three launches replayed in the step's order.  `traffic` = HBM bytes per launch
    (PMC: FETCH_SIZE x 2 + WRITE_SIZE, separate passes) and `mfma_busy_pct` from the committed counters of the same kernel."""
    seq = measure_update_sequence(agent) if seq is None else seq
    if not seq:
        return None
    e = seq[0]
    total = seq[-1]["us_per_launch"] if len(seq) > 3 else None
    rec, src = update_pmc_record(repo, e["entry_point"])
    out = {"bound": "mfma", "kernel": (rec or {}).get("kernel", "ag::split_gemm_kernel<true,...>"), "entry_point": e["entry_point"],
           "achieved": e["achieved"], "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": e["frac"],
           "traffic": (rec or {}).get("traffic_bytes_per_launch"), "traffic_source": src,
           "mfma_busy_pct": (rec or {}).get("mfma_busy_pct"),
           "us_per_launch": e["us_per_launch"], "algo_flops_per_launch": 6.0 * e["matrix_core_gflop_f32_equivalent"] * 1e9,
           "f32_equivalent_tflops": e["f32_equivalent_tflops"], "f32_equivalent_frac_of_f32_mfma_peak": e["f32_equivalent_tflops"] / FP32_MFMA_PEAK_TFLOPS,
           "hbm_algo_bytes_per_launch": e["hbm"]["algo_bytes"], "launches_per_epoch": optimizer_steps_per_epoch,
           "three_launch_sum_us": total, "kernel_source_sha": update_source_sha()}
    return out


def pmc_field(repo, key, kernel, field):
    """One extra field (e.g. `valu_busy_pct`) of the env-kernel PMC record bench.py quotes traffic from; None when stale / absent."""
    import glob
    import json
    import os
    sha = env_kernel_source_sha()
    for path in sorted(glob.glob(os.path.join(repo, "profiles", "r*_env_kernel_pmc.json")), reverse=True):
        rec = json.load(open(path)).get(key)
        if rec and rec.get("kernel") == kernel and rec.get("source_sha") == sha:
            return rec.get(field)
    return None


def planning_source_sha():
    """Provenance key of the Planning render kernel (sources it is compiled from), as env_kernel_source_sha."""
    import hashlib
    import os
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
    h = hashlib.sha256()
    for f in ("planning_kernel.hip", "planning_math.hpp", "env_math.hpp", "kernel_args.hpp"):
        with open(os.path.join(here, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measure_planning_render(env, steps=16):
    """The depth camera of a Planning handle (planning_render_kernel: ray-cast + the reference's post-processing in LDS, one
    212 x 120 float32 image per env written once, customized.py:386-435) timed in place: `steps` consecutive env steps with HIP
    events around each; the camera runs every 4th step, so render time = median(step with render) - median(step without).
    Algorithmic bytes: the image, 101 760 B per env and render - the kernel is vector-ALU bound (ray tests),
    the HBM fraction says how far from a memory problem it is."""
    import statistics
    n = env.num_envs
    a = torch.zeros(n, 4, device=env.device)
    a[:, 3] = -0.69
    for _ in range(4):
        env.step(a)
    s = torch.cuda.current_stream(env.device)
    with_r, without = [], []
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        env.step(a)
        e1.record(s)
        e1.synchronize()
        (with_r if env.last_step_rendered() else without).append(e0.elapsed_time(e1) * 1e3)
    if not with_r or not without:
        return None
    us = statistics.median(with_r) - statistics.median(without)
    nbytes = float(n) * 212 * 120 * 4
    return {"us_per_launch": us, "renders_timed": len(with_r), "algo_bytes": nbytes, "achieved": nbytes / us / 1e3,
            "us_step_with_render": statistics.median(with_r), "us_step_without": statistics.median(without)}


def planning_render_roofline(env, repo):
    m = measure_planning_render(env)
    if m is None:
        return None
    import glob
    import json
    import os
    traffic, src, valu = None, "no PMC record for this kernel", None
    sha = planning_source_sha()
    for path in sorted(glob.glob(os.path.join(repo, "profiles", "r*_planning_render_pmc.json")), reverse=True):
        rec = json.load(open(path))
        if rec.get("envs") != env.num_envs:
            continue
        if rec.get("source_sha") != sha:
            src = f"stale: {os.path.basename(path)} was measured on kernel sources {rec.get('source_sha')}, this build is {sha}"
            continue
        traffic, src, valu = rec["traffic_bytes_per_launch"], rec["source"], rec.get("valu_busy_pct")
        break
    return {"bound": "hbm", "kernel": "ag::planning_render_kernel<0>", "entry_point": "ag_step (Planning; the camera launch of every 4th step)",
            "achieved": m["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": m["achieved"] / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": src, "valu_busy_pct": valu, "us_per_launch": m["us_per_launch"],
            "algo_bytes_per_env": 212 * 120 * 4, "envs_per_launch": env.num_envs, "renders_timed": m["renders_timed"],
            "note": "ray-casting kernel: vector-ALU bound, the image crosses HBM once (the post-processing runs in LDS)",
            "kernel_source_sha": sha}


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
def measure_update_sequence(agent, iters=30):
    """The launches that dominate an optimizer step, EXACTLY as FusedMLPStep.step issued them (its own closures, its own live
    buffers), replayed in the step's order - forward (+ loss + head backward), weight gradient, dX (+ first-layer backward) -
    with HIP events around every launch, so that each kernel finds the caches as its predecessor leaves them.  Per kernel BOTH
    rooflines: matrix-core FLOPs at 6 bf16 MFMAs per f32 product (incl. the first layer / recomputation products the launch
    carries) against the dense bf16 peak, and algorithmic HBM bytes against 8 TB/s."""
    import statistics
    fs = getattr(agent, "_fused_step", None)
    if fs is None or not all(k in fs.last_launches for k in ("forward", "wgrad", "dx")):
        return []
    M, A1 = fs.M, fs.A + 1
    D = fs.layers[0][0].shape[1]
    order = [fs.last_launches[k] for k in ("forward", "wgrad", "dx")]
    s = torch.cuda.current_stream()
    for _ in range(3):
        for _, fn in order:
            fn()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(iters)]
    for it in range(iters):
        ev[it][0].record(s)
        for j, (_, fn) in enumerate(order):
            fn()
            ev[it][j + 1].record(s)
    torch.cuda.synchronize()
    us = [statistics.median(ev[it][j].elapsed_time(ev[it][j + 1]) * 1e3 for it in range(iters)) for j in range(3)]
    act, io = 4.0 * M * 256, 4.0 * M * D
    main = 2.0 * M * 256 * 256
    small = 2.0 * M * 32 * 256                 # one K = 32 (or K = rows x 32 columns) product: first layer, its recomputation, dW1
    rc, fin = bool(getattr(fs, "recompute_h1", False)), bool(getattr(fs, "fuse_gemm_input", False))
    S = fs.wgrad_partials[1].shape[0] if len(fs.wgrad_partials) > 1 else 0
    tiles = fs.wgrad_partials[0].shape[0]
    flops = [main + (small if fin else 0.0), main + (small if rc else 0.0), main + small + (small if rc else 0.0)]
    nbytes = [(2 * io if fin else act) + act + (0.0 if (rc or not fin) else act),        # obs in, xn out (or h1 in), dz out, h1 out
              act + (io if rc else act) + 4.0 * S * 65536,                                 # dz in, x / h1 in, slice partials out
              act + io + (0.0 if rc else act) + 4.0 * tiles * 256 * (D + 1)]               # dz in, x in, h1 in, tile partials out
    what = ["forward of both hidden layers + ELU + heads + PPO loss + head backward" if fin else
            "forward of the last hidden layer + ELU + heads + PPO loss + head backward",
            "weight gradient dW2 = dz2^T h1" + (" (h1 produced on chip from the network inputs)" if rc else ""),
            "dX of layer 2 + ELU' + dW1 / db1 in its epilogue" + (" (h1 recomputed on chip)" if rc else "")]
    out = []
    for (name, _), u, fl, nb, w in zip(order, us, flops, nbytes, what):
        out.append({"kernel": f"{name}: {w} - as the step runs it (timed in the step's launch order, {iters} rounds, median)",
                    "entry_point": name.split(" ")[0], "in_step": True, "us_per_launch": u, "bound": "mfma",
                    "achieved": 6.0 * fl / u / 1e6, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": 6.0 * fl / u / 1e6 / BF16_MFMA_PEAK_TFLOPS, "f32_equivalent_tflops": fl / u / 1e6,
                    "matrix_core_gflop_f32_equivalent": fl / 1e9,
                    "hbm": {"algo_bytes": nb, "achieved": nb / u / 1e3, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": nb / u / 1e3 / HBM_PEAK_GBPS}})
    out.append({"kernel": "sum of the three launches above", "in_step": True, "us_per_launch": sum(us),
                "hbm": {"algo_bytes": sum(nbytes), "note": "algorithmic HBM bytes of the three launches per optimizer step"}})
    return out


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
    # the weight gradient of the 256 x 256 layer: the library's split-K batched GEMM (f32 MFMA; what use_split_wgrad: false runs)
    S = 64
    if M % S == 0:
        lparts = torch.empty(S, C, K, dtype=torch.float32, device=x.device)
        dzv, xv = fs.dz[:M * C].view(S, M // S, C).transpose(1, 2), x.view(S, M // S, K)
        us = _time_us(lambda: torch.bmm(dzv, xv, out=lparts), iters)
        out.append({"kernel": f"library f32 split-K weight gradient [{C}x{M}]x[{M}x{K}] ({S} slices) - for reference"
                              + ("" if getattr(fs, "split_wgrad", None) else " (what the step runs)"), "bound": "mfma",
                    "achieved": flops / us / 1e6, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": flops / us / 1e6 / FP32_MFMA_PEAK_TFLOPS, "us_per_launch": us,
                    "note": "f32-input MFMA (1/16 of the bf16 rate); reads both operands twice (profiles/r02_update_kernels_pmc.md)"})
        del lparts
    if getattr(fs, "split_wgrad", None):
        li = max(fs.split_wgrad)
        wp = fs.wgrad_partials[li]
        dzm = fs.dz[:M * C].view(M, C)
        us = _time_us(lambda: N.check(lib.ag_split_wgrad(dzm.data_ptr(), x.data_ptr(), wp.data_ptr(), M, C, K, wp.shape[0], st),
                                      "ag_split_wgrad"), iters)
        out.append({"kernel": f"ag_split_wgrad dW[{C}x{K}] = dZ^T X over {M} rows ({wp.shape[0]} row slices, 6 bf16 MFMAs per f32 "
                              f"product, f32-accurate), stored-h1 form"
                              + (" - for reference (the step runs ag_split_wgrad_input)" if getattr(fs, "recompute_h1", False) else ""),
                    "bound": "mfma", "achieved": 6.0 * flops / us / 1e6,
                    "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": 6.0 * flops / us / 1e6 / BF16_MFMA_PEAK_TFLOPS,
                    "us_per_launch": us, "f32_equivalent_tflops": flops / us / 1e6,
                    "note": "each operand read from HBM once; slice partials summed by ag_sum_rows_multi (fixed order)"})
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
            mfma("ag_split_gemm_elu_heads (forward of the last hidden layer + ELU + heads; the rollout's form without the chain "
                 "kernel) - for reference", us,
                 flops + 2.0 * M * C * A1, "replaces ag_split_gemm + ag_elu_heads (one pass over z less)")
        if getattr(fs, "fuse_gemm_loss", False):
            lrows = lib.ag_split_gemm_loss_rows()
            tiles = M // lrows
            f = dict(dtype=torch.float32, device=x.device)
            zeros = lambda *s: torch.zeros(*s, **f)
            Lp = N.AgLossEpilogue()
            Lp.struct_size = ctypes.sizeof(N.AgLossEpilogue)
            keep = {"act": zeros(M, fs.A), "nlp": zeros(M), "adv": zeros(M), "ret": zeros(M), "val": zeros(M), "mu": zeros(M, fs.A),
                    "sig": torch.ones(M, fs.A, **f), "lp": torch.empty(tiles, lib.ag_ppo_loss_num_sums(), **f),
                    "dwh": torch.empty(tiles, A1, C, **f), "db": torch.empty(tiles, C, **f)}
            Lp.logstd_dev = agent.model.logstd.data_ptr()
            Lp.actions_dev, Lp.old_neglogp_dev, Lp.advantages_dev = keep["act"].data_ptr(), keep["nlp"].data_ptr(), keep["adv"].data_ptr()
            Lp.returns_dev, Lp.old_values_dev = keep["ret"].data_ptr(), keep["val"].data_ptr()
            Lp.old_mu_dev, Lp.old_sigma_dev, Lp.new_mu_dev, Lp.new_sigma_dev = keep["mu"].data_ptr(), keep["sig"].data_ptr(), None, None
            Lp.heads_dev = None
            Lp.loss_partials_dev, Lp.dwh_partials_dev, Lp.db_partials_dev = keep["lp"].data_ptr(), keep["dwh"].data_ptr(), keep["db"].data_ptr()
            Lp.e_clip, Lp.critic_coef, Lp.bounds_loss_coef, Lp.clip_value, Lp.bound_type = 0.2, 2.0, 1e-4, 0, 1
            us = _time_us(lambda: sg.forward_loss_heads_bwd(x, fs.dz[:M * C].view(M, C), b, agent.heads_w, agent.heads_b, Lp), iters)
            mfma("ag_split_gemm_loss_heads_bwd (forward of the last hidden layer + ELU + heads + PPO loss + head backward; h1 read "
                 "from HBM)" + (" - for reference (the step runs the fused-first-layer form, first entries)"
                                if getattr(fs, "fuse_gemm_input", False) else ""), us, flops + 4.0 * M * C * A1,
                 "replaces ag_split_gemm_elu_heads + ag_ppo_loss + ag_heads_bwd_elu_wgrad: z, heads and d_heads never touch HBM")
            del keep
        if getattr(fs, "fuse_gemm_input_wgrad", False):
            D0 = fs.layers[0][0].shape[1]
            us = _time_us(lambda: sg.backward_input_wgrad(h, fs.h[0], fs.xn, fs.wgrad_partials[0], fs.bias_partials[0]), iters)
            mfma("ag_split_gemm_input_wgrad (dX of layer 2 + ELU' + dW1 / db1 on the matrix cores, stored-h1 form)"
                 + (" - for reference (the step runs ag_split_gemm_input_wgrad_recompute)" if getattr(fs, "recompute_h1", False) else ""), us,
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
    # partial buffers of ITS block count (the step's own are sized for whichever kernel the step runs: with the loss in the
    # GEMM epilogue they hold one row per 256-row GEMM tile, this kernel writes one per 128 rows)
    hb_blocks = (M + lib.ag_wgrad_rows_per_block(0) - 1) // lib.ag_wgrad_rows_per_block(0)
    parts = torch.empty(hb_blocks, C, dtype=torch.float32, device=x.device)
    hparts = torch.empty(hb_blocks, A1, C, dtype=torch.float32, device=x.device)
    us = _time_us(lambda: lib.ag_heads_bwd_elu_wgrad(fs.d_heads.data_ptr(), agent.heads_w.data_ptr(), h.data_ptr(),
                                                      scratch.data_ptr(), parts.data_ptr(), hparts.data_ptr(),
                                                      M, C, A1, 1, None, st), iters)
    hbm("ag_heads_bwd_elu_wgrad (head dX + ELU' + head wgrad)" + (" - folded into the forward GEMM's epilogue in the step, shown for "
        "reference" if getattr(fs, "fuse_gemm_loss", False) else ""), us, 4.0 * M * (2 * C + A1))
    del parts, hparts
    if fs.fuse_input_wgrad and lib.ag_input_wgrad_rows(fs.layers[0][0].shape[1]) > 0:
        D = fs.layers[0][0].shape[1]
        C0 = fs.layers[0][0].shape[0]
        iparts = fs.bias_partials[0]
        us = _time_us(lambda: lib.ag_elu_bwd_input_wgrad(fs.dh.data_ptr(), fs.h[0].data_ptr(), fs.xn.data_ptr(),
                                                          fs.wgrad_partials[0].data_ptr(), iparts.data_ptr(), M, C0, D, st), iters)
        hbm("ag_elu_bwd_input_wgrad (ELU' + first-layer wgrad)" + (" - folded into the dX GEMM's epilogue in the step, shown for reference"
            if getattr(fs, "fuse_gemm_input_wgrad", False) else ""), us, 4.0 * M * (2 * C0 + D))
    return out
