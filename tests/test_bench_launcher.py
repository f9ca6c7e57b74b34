"""bench.py as the driver launches it: `python bench.py --gpus N` with no WORLD_SIZE in the environment must become N ranks by
itself, run the data-parallel job (reference: one process per GPU, lib/agent/a2c_base.py:109-123; one gradient all-reduce per
optimizer step, :293-309) and print ONE rank-0 JSON line.  Here: world 2 over gloo on the CPU test double of the env."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=420):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, cwd=REPO, env=env, capture_output=True,
                       text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def test_self_launch_two_ranks_gloo():
    r, lines = _run(["--gpus", "2", "--device", "cpu", "--agent", "tests._stub_bench_agent:StubAgent", "--envs", "16",
                     "--steps", "2", "--warmup", "1", "--minibatches", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout          # exactly one JSON line, from rank 0
    assert len(lines[0]) <= 6000, len(lines[0])      # the driver's record keeps a bounded tail of stdout (round 5: 20 KB -> parsed null)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_envs"] == 32 and out["config"]["parallelism"] == "dp2"
    assert out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak" and out["higher_is_better"] is True
    # whole-job value: both ranks' env steps over the max-over-ranks time
    assert abs(out["value"] - 2 * 16 * out["config"]["horizon_length"] * 2 / (out["ms_per_step"] * 2 / 1e3)) < 1e-6 * out["value"]
    rc = out["rccl"]
    assert rc["ranks_seen"] == 2 and rc["ranks_counted_by_allreduce"] == 2 and rc["backend"] == "gloo"
    assert rc["allreduce_us"] > 0 and rc["bytes"] > 0 and rc["per_epoch"] == 5 * 2
    # counted, not derived: 10 gradient all-reduces per epoch; the input normaliser's moments in the first mini-epoch's two
    # minibatches + the value normaliser's two updates (values, returns) per epoch
    cpe = rc["collectives_per_epoch"]
    assert cpe["gradient"]["calls"] == 10 and cpe["gradient"]["bytes"] == 10 * rc["bytes"]
    assert cpe["normaliser_moments"]["calls"] == 2 + 2
    assert set(cpe) == {"gradient", "normaliser_moments"}
    assert rc["launch"]["attempt"] == 1 and rc["launch"]["earlier"] == [] and rc["minibatch_hip_graphs"] is False
    assert out["phases"]["finite"]
    assert "cpu_baseline" not in out and "shipped_ratio" not in out      # N == 1 legs stay out of the N > 1 line


def test_too_few_devices_is_one_json_error_line():
    """A 1-GPU (here: 0-GPU) box asked for 2 GPUs: exit code 0 and one parseable line that says why nothing was measured."""
    r, lines = _run(["--gpus", "2", "--steps", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["value"] is None and out["n_gpus"] == 2 and "needs 2 devices" in out["error"]


def test_mismatched_world_size_is_refused():
    r, _ = _run(["--gpus", "2", "--device", "cpu"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_self_launch_eight_ranks_gloo():
    """The shape of the driver's N = 8 run (one process per GPU of one node), on the CPU test double."""
    r, lines = _run(["--gpus", "8", "--device", "cpu", "--agent", "tests._stub_bench_agent:StubAgent", "--envs", "8",
                     "--steps", "1", "--warmup", "1", "--minibatches", "2"], env_extra={"OMP_NUM_THREADS": "1"}, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["global_envs"] == 64 and out["config"]["parallelism"] == "dp8"
    rc = out["rccl"]
    assert rc["ranks_seen"] == 8 and rc["ranks_counted_by_allreduce"] == 8
    assert rc["collectives_per_epoch"]["gradient"]["calls"] == 10
    assert abs(out["value"] - 8 * 8 * out["config"]["horizon_length"] / (out["ms_per_step"] / 1e3)) < 1e-6 * out["value"]
    assert out["phases"]["finite"]


def test_failing_collective_leaves_one_json_line_with_what_was_seen():
    r, lines = _run(["--gpus", "2", "--device", "cpu", "--agent", "tests._stub_bench_agent:BrokenCollectiveAgent", "--envs", "8",
                     "--steps", "1", "--warmup", "0", "--minibatches", "2"])
    assert r.returncode != 0
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["value"] is None and out["n_gpus"] == 2
    att = out["launch_attempts"]
    assert len(att) == 1 and att[0]["exit_code"] != 0                  # cpu: one attempt (no IPC setting to flip)
    assert "hipIpcGetMemHandle" in att[0]["error"] and "initial parameter broadcast" in att[0]["error"]
    assert out["rccl"]["ranks_seen"] == 2 and out["rccl"]["backend"] == "gloo"      # what the group reported before it died


def test_committed_pmc_record_was_measured_on_these_kernel_sources():
    """bench.py quotes `roofline.traffic` only from a PMC record whose provenance key equals the key of the env-step kernel's sources
    and compile flags in the tree (`env_kernel_source_sha`): the newest committed record must be current, or the driver's line
    carries `traffic: null`.  (Re-measure with tools/gpu_pmc_env.sh after touching step_kernel.hip / env_math.hpp / kernel_args.hpp /
    rollout_math.hpp or the step units' flags in build.py.)"""
    import os
    from airgym_amd.utils.kernel_bench import kernel_name, pmc_traffic
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for key, kernel in (("hovering_rate_multi24", kernel_name("hovering", "rate", False)),
                        ("hovering_rate_fused", kernel_name("hovering", "rate", True)),
                        ("hovering_rate", kernel_name("hovering", "rate", False, True))):
        traffic, source = pmc_traffic(repo, key, kernel)
        assert traffic is not None and traffic > 0, (key, source)


def _fat_record():
    """A full bench record of the size round 5 printed (prose notes, per-kernel lists, nested side configurations)."""
    prose = "x" * 400
    kern = {"bound": "hbm", "kernel": "ag::step_kernel_ws2<0,3,true>", "entry_point": prose, "us_per_launch": 10.174405116301317,
            "achieved": 2415.472916507385, "peak": 8000.0, "unit": "GB/s", "frac": 0.3019341145634231, "traffic": 27096064,
            "traffic_source": prose, "algo_bytes_per_env_step": 375, "note": prose, "launches_timed": 2496}
    ek = {"in_loop": kern, "single_step": dict(kern), "multi_step": dict(kern, steps_per_launch=24, valu_busy_pct=87.0),
          "copy_ceiling_gbps": 5132.929068638738}
    return {
        "metric": "env_steps_per_sec_hovering_65536_envs_per_gpu", "value": 63208714.26767838, "unit": "env-steps/s", "n_gpus": 1,
        "steps": 20, "warmup": 5, "ms_per_step": 24.883657549798954, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "hovering_ctbr_ppo_epoch", "task": "hovering", "ctl_mode": "rate", "envs_per_gpu": 65536,
                   "global_envs": 65536, "horizon_length": 24, "mini_epochs": 5, "minibatch_size": 196608, "policy": "MLP(256,256)",
                   "parallelism": "dp1", "hidden_layer_gemm": prose, "paths": {"update": {f"layer{i}": prose for i in range(4)}}},
        "phases": {"rollout_host_enqueue_s": 0.005, "update_s": 0.49, "last_kl": 0.013, "finite": True, "final_lr": 3e-3},
        "roofline": {"bound": "mfma", "kernel": "ag::split_gemm_kernel<true,5,0,4,true,18,true>", "entry_point": "ag_split_gemm_input_loss_heads_bwd",
                     "achieved": 748.1, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.2992, "traffic": 265000000, "traffic_source": prose,
                     "us_per_launch": 233.6, "launches_per_epoch": 40, "mfma_busy_pct": 32.6},
        "env_kernels": ek, "env_only": {"value": 1.5e10, "unit": "env-steps/s", "note": prose},
        "update_kernels": [dict(kern, kernel=prose) for _ in range(14)],
        "side_configs": {"tracking_lv": {"value": 5.4e7, "ms_per_step": 28.9, "dtype": "f32", "config": {"paths": prose}, "env_kernels": ek},
                         "hovering_bf16": {"value": 1.09e8, "ms_per_step": 14.4, "dtype": "bf16", "update_kernels": [kern] * 3},
                         "planning_cnn_16384": {"value": 3.8e5, "ms_per_step": 1026.0, "dtype": "f32", "rollout_ms": 120.0, "update_ms": 900.0,
                                                "roofline": dict(kern)}},
        "shipped_ratio": {"value": 3.6e7, "ms_per_step": 43.6, "minibatch_size": 32768, "note": prose},
        "cpu_baseline": {"value": 1864889.8, "unit": "env-steps/s", "cores": 8, "kind": "port", "host_cores": 256, "sample": prose,
                         "thread_sweep": {str(t): 1e6 for t in (1, 4, 8, 16, 32, 64)},
                         "config0": {"value": 30570.9, "envs": 64, "threads": 1, "sample": prose}}}


def test_stdout_line_is_compact_and_complete():
    """the line the driver parses stays under 4 KB whatever the detail record holds, and still carries the
    contract's keys, `roofline` (bound / achieved / peak / unit / frac / traffic) and `cpu_baseline` (value / unit / cores / kind /
    sample).  Everything else lives in bench_detail.json."""
    sys.path.insert(0, REPO)
    import bench
    rec = _fat_record()
    assert len(json.dumps(rec)) > 15000
    line = bench.compact_line(rec, detail_path="gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) <= bench.LINE_LIMIT <= 4096, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == rec["value"] and line["ms_per_step"] == rec["ms_per_step"]        # full precision where the driver checks
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    assert line["config"]["workload"] == "hovering_ctbr_ppo_epoch" and "paths" not in line["config"]
    assert line["env_kernels"]["in_loop"]["frac"] == 0.3019 and line["side"]["planning_cnn_16384"]["value"] == 3.8e5
    assert not any(isinstance(v, str) and len(v) > 130 for v in _walk(line)), "prose belongs in the detail file"
    # a record that is too fat even when compacted loses optional blocks, never the contract's keys
    rec["rccl"] = {"ranks_seen": 8, "collectives_per_epoch": {f"tag{i}": {"calls": 1.0, "bytes": 2.0} for i in range(120)}}
    small = bench.compact_line(rec)
    assert len(json.dumps(small)) <= bench.LINE_LIMIT and "roofline" in small and "cpu_baseline" in small and "value" in small


def _walk(o):
    if isinstance(o, dict):
        for v in o.values():
            yield from _walk(v)
    elif isinstance(o, list):
        for v in o:
            yield from _walk(v)
    else:
        yield o
