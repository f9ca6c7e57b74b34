"""Does the fast path LEARN?  Regression for the headline configuration (Hovering / CTBR, 65 536 envs, MLP(256,256), 196 608-sample
minibatches; split-bf16 GEMMs, fused epilogues, fused rollout step, hipGraph rollout): the mean episode reward must reach 2 000
within 120 epochs on at least 4 of 5 seeds.  The 5-seed x 8-arm study this bar comes from is profiles/r03_seed_study.md: there
all five seeds of this arm reach 6 700-7 700 by epoch 120 (random policy: ~20; hover for the whole 2 400-step episode: ~8 000)."""
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
    best = []
    for seed in range(5):
        out = run(f"headline seed {seed}", 65536, 8, 120, 10, seed=seed)
        best.append(max((c["reward"] or 0.0) for c in out["curve"]))
        assert all(c["kl"] == c["kl"] and c["c_loss"] == c["c_loss"] for c in out["curve"]), "NaN in the losses"
    assert sum(b >= 2000.0 for b in best) >= 4, best


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
