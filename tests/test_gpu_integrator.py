"""Analytic checks of the rigid-body integrator inside the HIP step kernel - INDEPENDENT of oracle/rigid_body.py.

The reference integrates with PhysX (closed), so nothing in the reference can pin this piece ("parity
unpinned").  What can be pinned is physics: closed-form solutions and conserved quantities of a free rigid body with the
X152b mass properties (airgym/assets/robots/X152b/model.urdf:19-24,36-39,86-105: M = 0.601 kg,
I = diag(0.0400591785, 0.0400591785, 0.0400964156)), the thrust / reaction-torque constants of hovering.py:256-275
(9.59 N and 0.2 N m per unit command, arm 0.05374 m) and the speed caps of airgym/assets/__init__.py:30-35.

`prop` mode (SRT, hovering.py:251-252) makes the rotor commands equal to the actions, so the applied wrench is known
exactly.  Expected values are computed in float64 numpy here; tolerances are ~5x what float32 RK4 at dt = 0.01 achieves.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

M = 0.601
I = np.array([0.0400591785, 0.0400591785, 0.0400964156])
G = 9.81
KF, KM, ARM = 9.59, 0.2, 0.05374
DT = 0.01


@pytest.fixture(scope="module")
def Handle():
    from airgym_amd.hip_env import HipEnvHandle
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return HipEnvHandle


def _run(Handle, rs, cmd, steps, max_episode_length=0):
    """Advance a prop-mode Hovering handle from root states `rs` [n,13] under constant rotor commands `cmd` [n,4];
    returns the trajectory [steps+1, n, 13] (float64 numpy) and the number of terminations."""
    n = rs.shape[0]
    env = Handle("hovering", "prop", n, seed=0, obs_noise=False, max_episode_length=max_episode_length)
    env.set_state(root_states=torch.as_tensor(rs, dtype=torch.float32), progress=torch.zeros(n, dtype=torch.int32),
                  pre_actions=torch.zeros(n, 4), was_reset=torch.zeros(n, dtype=torch.int32),
                  ctl_state=torch.zeros(n, 12))
    act = torch.as_tensor(cmd, dtype=torch.float32).cuda().contiguous()
    out = [env.get_state()["root_states"].cpu().double().numpy()]
    resets = 0
    for _ in range(steps):
        env.step(act)
        resets += int(env.reset_buf.sum())
        out.append(env.get_state()["root_states"].cpu().double().numpy())
    env.close()
    return np.stack(out), resets


def _qrot(q, v, inverse=False):
    qv, w = q[..., :3], q[..., 3:4]
    t2 = 2.0 * np.cross(qv, v)
    return v + (-w if inverse else w) * t2 + np.cross(qv, t2)


def test_free_fall_is_a_parabola(Handle):
    """Zero rotor command: a = (0, 0, -9.81); RK4 is exact for quadratics, so only float32 rounding remains."""
    rs = np.zeros((4, 13)); rs[:, 6] = 1.0; rs[:, 2] = 1.5
    rs[:, 7:10] = [[0.3, -0.2, 0.5], [0.0, 0.0, 0.0], [1.0, 1.0, -0.5], [-0.7, 0.2, 1.0]]
    tr, resets = _run(Handle, rs, np.zeros((4, 4)), 55)
    assert resets == 0
    t = (np.arange(56) * DT)[:, None]
    np.testing.assert_allclose(tr[:, :, 2], rs[:, 2] + rs[:, 9] * t - 0.5 * G * t ** 2, rtol=0, atol=3e-6)
    np.testing.assert_allclose(tr[:, :, 9], rs[:, 9] - G * t, rtol=0, atol=1e-5)
    for k in (0, 1):
        np.testing.assert_allclose(tr[:, :, k], rs[:, k] + rs[:, 7 + k] * t, rtol=0, atol=2e-6)
        np.testing.assert_allclose(tr[:, :, 7 + k], np.broadcast_to(rs[:, 7 + k], tr[:, :, 0].shape), rtol=0, atol=1e-7)
    np.testing.assert_allclose(tr[:, :, 3:7], np.broadcast_to(rs[:, 3:7], tr[:, :, 3:7].shape), rtol=0, atol=1e-7)
    assert np.abs(tr[:, :, 10:13]).max() == 0.0


def test_constant_yaw_torque_closed_form(Handle):
    """Commands (c, c, c+d, c+d): pure reaction torque 0.2*2d about body z (hovering.py:270-275), no roll / pitch moment.
    From rest: w_z = a t, yaw = a t^2 / 2 with a = 0.4 d / Izz, q = (0, 0, sin(yaw/2), cos(yaw/2)); thrust stays vertical:
    v_z = (9.59 (4c + 2d) / M - g) t."""
    d = np.array([0.02, 0.05, -0.04]); c = 0.14
    cmd = np.stack([np.full(3, c), np.full(3, c), c + d, c + d], 1)
    rs = np.zeros((3, 13)); rs[:, 6] = 1.0
    tr, resets = _run(Handle, rs, cmd, 100)
    assert resets == 0
    t = (np.arange(101) * DT)[:, None]
    a = KM * 2 * d / I[2]
    yaw = 0.5 * a * t ** 2
    az = KF * (4 * c + 2 * d) / M - G
    np.testing.assert_allclose(tr[:, :, 12], a * t, rtol=0, atol=2e-6)
    np.testing.assert_allclose(tr[:, :, 5], np.sin(yaw / 2), rtol=0, atol=1e-6)
    np.testing.assert_allclose(tr[:, :, 6], np.cos(yaw / 2), rtol=0, atol=1e-6)
    np.testing.assert_allclose(tr[:, :, 9], az * t, rtol=0, atol=1e-5)
    np.testing.assert_allclose(tr[:, :, 2], 0.5 * az * t ** 2, rtol=0, atol=5e-6)
    assert np.abs(tr[:, :, [0, 1, 3, 4, 7, 8, 10, 11]]).max() < 1e-7


def test_constant_roll_torque_closed_form(Handle):
    """Commands (c-d, c+d, c+d, c-d): pure moment 4 d * 9.59 * 0.05374 about body x (rotor arms, model.urdf:86-105), a
    principal axis, so there is no gyroscopic coupling: w_x = a t, q = (sin(a t^2 / 4), 0, 0, cos(a t^2 / 4))."""
    d = np.array([0.01, 0.03, -0.02]); c = 0.14
    cmd = np.stack([c - d, c + d, c + d, c - d], 1)
    rs = np.zeros((3, 13)); rs[:, 6] = 1.0
    tr, resets = _run(Handle, rs, cmd, 100)
    assert resets == 0
    t = (np.arange(101) * DT)[:, None]
    a = ARM * KF * 4 * d / I[0]
    np.testing.assert_allclose(tr[:, :, 10], a * t, rtol=0, atol=6e-6)
    np.testing.assert_allclose(tr[:, :, 3], np.sin(0.25 * a * t ** 2), rtol=0, atol=1e-6)
    np.testing.assert_allclose(tr[:, :, 6], np.cos(0.25 * a * t ** 2), rtol=0, atol=1e-6)
    assert np.abs(tr[:, :, [4, 5, 11, 12]]).max() < 1e-7
    # the thrust tilts with the body: lateral acceleration -(F/M) sin(roll), so y moves opposite to the roll sign
    assert (np.sign(tr[-1, :, 1]) == -np.sign(d)).all()


def test_torque_free_tumbling_conserves_momentum_and_energy(Handle):
    """Zero command, 3-D initial body rates: angular momentum L = R (I w_b) is constant IN THE WORLD FRAME and the rotational
    energy w_b . I w_b / 2 is constant; the centre of mass follows the free-fall parabola regardless of the tumbling."""
    wb = np.array([[2.0, 1.0, 3.0], [-1.5, 2.5, 0.5], [0.3, -2.0, -2.0], [3.0, 0.0, 0.1]])
    rs = np.zeros((4, 13)); rs[:, 6] = 1.0; rs[:, 2] = 1.9; rs[:, 10:13] = wb
    tr, resets = _run(Handle, rs, np.zeros((4, 4)), 30)
    assert resets == 0
    q, w = tr[:, :, 3:7], tr[:, :, 10:13]
    w_b = _qrot(q, w, inverse=True)
    L = _qrot(q, w_b * I)
    E = 0.5 * (w_b * w_b * I).sum(-1)
    assert (np.linalg.norm(L - L[0], axis=-1) / np.linalg.norm(L[0], axis=-1)).max() < 5e-6
    assert (np.abs(E - E[0]) / E[0]).max() < 8e-6
    np.testing.assert_allclose(np.linalg.norm(q, axis=-1), 1.0, rtol=0, atol=3e-7)
    t = (np.arange(31) * DT)[:, None]
    np.testing.assert_allclose(tr[:, :, 2], np.broadcast_to(1.9 - 0.5 * G * t ** 2, tr[:, :, 2].shape), rtol=0, atol=3e-6)
    assert (1 - 2 * (q[..., 0] ** 2 + q[..., 1] ** 2)).min() > 0.3          # the body really tilted


def test_symmetric_top_precession_invariants_full_episode(Handle):
    """Spinning about (almost) the symmetry axis with the angular momentum vertical and the thrust balancing gravity
    along it: over a whole 24 s episode (2 398 steps, ~19 revolutions) |L|, L's world direction, the rotational energy,
    the body-z rate and |w_xy| are invariant; the vehicle stays inside the termination box, so no reset interferes."""
    wb = np.array([[0.1, 0.0, 5.0], [0.06, -0.08, 4.0], [0.0, 0.12, -6.0], [0.05, 0.05, 3.0]])
    Lb = wb * I
    s = np.sign(Lb[:, 2:3])
    a = s * Lb / np.linalg.norm(Lb, axis=1, keepdims=True)
    b = np.broadcast_to(np.array([0.0, 0.0, 1.0]), a.shape)
    q0 = np.concatenate((np.cross(a, b), 1.0 + (a * b).sum(-1, keepdims=True)), 1)
    q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    theta = np.arccos(a[:, 2])                                  # angle between L and the body z axis (thrust axis)
    rs = np.zeros((4, 13)); rs[:, 3:7] = q0; rs[:, 10:13] = _qrot(q0, wb)
    cmd = np.repeat((M * G / np.cos(theta) / KF / 4.0)[:, None], 4, 1)
    tr, resets = _run(Handle, rs, cmd, 2398)
    assert resets == 0
    q, w = tr[:, :, 3:7], tr[:, :, 10:13]
    w_b = _qrot(q, w, inverse=True)
    L = _qrot(q, w_b * I)
    E = 0.5 * (w_b * w_b * I).sum(-1)
    assert (np.linalg.norm(L - L[0], axis=-1) / np.linalg.norm(L[0], axis=-1)).max() < 6e-5
    assert (np.abs(E - E[0]) / E[0]).max() < 1.2e-4
    assert np.abs(w_b[:, :, 2] - w_b[0, :, 2]).max() < 2.5e-4
    assert np.abs(np.linalg.norm(w_b[:, :, :2], axis=-1) - np.linalg.norm(w_b[0, :, :2], axis=-1)).max() < 1e-4
    np.testing.assert_allclose(np.linalg.norm(q, axis=-1), 1.0, rtol=0, atol=3e-7)
    assert np.abs(tr[:, :, 0:3]).max() < 2.0
    # L is vertical by construction and stays so
    assert np.abs(L[:, :, :2]).max() / np.abs(L[0, :, 2]).min() < 1e-4


def test_speed_caps(Handle):
    """airgym/assets/__init__.py:30-35 (max_linear_velocity / max_angular_velocity = 100): PhysX clamps the speeds after
    integration; direction is preserved."""
    rs = np.zeros((4, 13)); rs[:, 6] = 1.0
    rs[0, 7:10] = [0.0, 0.0, 150.0]
    rs[1, 7:10] = [90.0, 90.0, 20.0]
    rs[2, 10:13] = [0.0, 0.0, 200.0]
    rs[3, 7:10] = [30.0, 40.0, 0.0]; rs[3, 10:13] = [0.0, 0.0, 99.0]        # below both caps: untouched
    cmd = np.full((4, 4), M * G / KF / 4.0)                                   # hover thrust: no net force
    tr, resets = _run(Handle, rs, cmd, 1)
    v, w = tr[1, :, 7:10], tr[1, :, 10:13]
    assert abs(np.linalg.norm(v[0]) - 100.0) < 1e-3 and abs(np.linalg.norm(v[1]) - 100.0) < 1e-3
    np.testing.assert_allclose(v[1] / np.linalg.norm(v[1]), rs[1, 7:10] / np.linalg.norm(rs[1, 7:10]), atol=1e-4)
    assert abs(np.linalg.norm(w[2]) - 100.0) < 1e-3 and abs(w[2, 2] - 100.0) < 1e-3
    np.testing.assert_allclose(v[3], rs[3, 7:10], atol=2e-3)
    np.testing.assert_allclose(w[3], rs[3, 10:13], atol=1e-4)
