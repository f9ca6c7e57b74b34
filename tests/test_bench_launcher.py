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
