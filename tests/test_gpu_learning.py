"""Does the fast path LEARN?  Regression for the headline configuration (Hovering / CTBR, 65 536 envs, MLP(256,256), 196 608-sample
minibatches; split-bf16 GEMMs with h1 recomputed, fused epilogues incl. the loss, chain forward + fused rollout step, hipGraph
rollout).

What is asserted, and why not the reference's meter at epoch 200 (profiles/r05_collapse_trace.md): the meter averages the last
100 episodes that ENDED; once the policy keeps all 65 536 envs alive the only episodes that end between two time-limit waves are
the few that crash, so the meter drops to a few hundred while the population flies better than ever.  So:
  * default configuration: the meter must reach 2 000 within 120 epochs on >= 4 of 5 seeds (learning speed), and at epoch 200
    the mean raw reward per env-step over the WHOLE last rollout must be >= 3.0 on >= 4 of 5 seeds and >= 2.0 on all (a random
    policy collects ~1.3, a perfect hover ~3.6; the round-5 trace has 3.49 / 3.67 / 3.48 / 3.60 and one seed at 2.32 that is
    still inside the crash wave behind its time-limit wave) - a regression in any kernel of the path shows here;
  * the same five runs against three seeds of the PLAIN TORCH path (tools/learning_ab.py AUTOGRAD arm) at epoch 120: medians of the
    whole-population step reward within 0.25 of each other, the fused median not more than 0.1 below the torch arm's worst seed
    (profiles/r06_learning_ab.md: 15 seeds per arm, one distribution);
  * opt-in `max_lr: 1e-3` (the arm that keeps the recovery-from-reset skill through the reset-free phase): every env's FIRST
    episode from a fresh full reset under the final policy (tools/learning_curves.py evaluate_population, 65 536 episodes): the
    MEDIAN over five seeds must be >= 6 000 and every seed >= 3 000 (~8 000 = perfect).  Two measurements of this arm on builds that
    differ in float32 rounding only: 7 699 / 7 674 / 7 124 / 7 177 / 6 422 (the trace) and 5 254 / 6 492 / 7 009 / 4 752 / 7 770 (the final
    build) - a run is chaotic in its rounding, so the bar is on the median and the worst seed, not on a per-seed count."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_headline_configuration_learns_on_4_of_5_seeds():
    assert torch.cuda.is_available()
    sys.path.insert(0, REPO)
    from tools.learning_curves import run
    best, final, step_reward, sr120 = [], [], [], []
    for seed in range(5):
        out = run(f"headline seed {seed}", 65536, 8, 200, 10, seed=seed)
        best.append(max((c["reward"] or 0.0) for c in out["curve"] if c["epoch"] <= 120))
        sr120.append({c["epoch"]: c for c in out["curve"]}[120]["step_reward"])
        final.append(out["curve"][-1]["reward"] or 0.0)
        step_reward.append(out["final_step_reward"])
        assert out["curve"][-1]["epoch"] == 200
        assert all(c["kl"] == c["kl"] and c["c_loss"] == c["c_loss"] for c in out["curve"]), "NaN in the losses"
    print("meter: best by epoch 120:", best, "at epoch 200:", final, "| whole-population reward per env-step at epoch 200:", step_reward)
    assert sum(b >= 2000.0 for b in best) >= 4, best
    assert sum(r >= 3.0 for r in step_reward) >= 4 and all(r >= 2.0 for r in step_reward), step_reward
    # ... and a same-seed A/B against the plain torch path (autograd, library GEMMs, torch Adam, env.step per rollout step; round 6,
    # profiles/r06_learning_ab.md: over 15 seeds the two arms are one distribution - epoch-120 step reward 3.25 / 3.50 / 3.64 against
    # 3.11 / 3.54 / 3.65).  Single trajectories diverge chaotically, so the gate is on the arms' medians and the torch arm's spread.
    import statistics

    from tools.learning_ab import AUTOGRAD
    ref120 = []
    for seed in range(3):
        out = run(f"torch path seed {seed}", 65536, 8, 120, 10, seed=seed, extra=dict(AUTOGRAD))
        ref120.append({c["epoch"]: c for c in out["curve"]}[120]["step_reward"])
        assert out["curve"][-1]["epoch"] == 120
    print("whole-population reward per env-step at epoch 120: fused", sr120, "| torch path", ref120)
    mf, mr = statistics.median(sr120), statistics.median(ref120)
    assert abs(mf - mr) <= 0.25 and mf >= min(ref120) - 0.1, (sr120, ref120)


def test_opt_in_max_lr_arm_keeps_the_whole_population_flying_at_epoch_200():
    assert torch.cuda.is_available()
    sys.path.insert(0, REPO)
    from tools.learning_curves import run
    returns, lengths = [], []
    for seed in range(5):
        out = run(f"max_lr 1e-3 seed {seed}", 65536, 8, 200, 50, seed=seed, extra={"max_lr": 1e-3}, evaluate=True)
        returns.append(out["eval"]["eval_return"])
        lengths.append(out["eval"]["eval_length"])
        assert out["eval"]["eval_envs"] == 65536
    print("whole-population first-episode return at epoch 200 (max_lr 1e-3):", returns, "lengths:", lengths)
    assert sorted(returns)[2] >= 6000.0, returns
    assert all(r >= 3000.0 for r in returns), returns


def test_planning_cnn_policy_learns_on_the_hand_written_trunk():
    """Planning with the shipped YAML's trainable CNN (frame de-duplication, csrc/conv_kernels.hip, lib/network/fused_cnn.py) at
    2 048 envs: the mean episode reward must grow tenfold within 40 epochs (epoch 1: ~25, a 12-step episode; the three arms of
    profiles/r03_planning_learning_ab.md - this one, torch's conv2d, the reference's shape - all reach 500-730 by epoch 40)."""
    assert torch.cuda.is_available()
    sys.path.insert(0, REPO)
    from tools.planning_learning_ab import run
    out = run("hip_trunk", 2048, 40, 10, 0)
    assert out["dedup"]
    rewards = [c["reward"] for c in out["curve"] if c["reward"] is not None]
    assert all(c["kl"] == c["kl"] and c["c_loss"] == c["c_loss"] for c in out["curve"]), "NaN in the losses"
    assert max(rewards) >= 10.0 * rewards[0] and max(rewards) >= 250.0, rewards
