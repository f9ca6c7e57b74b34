"""oracle/rotations.py (pytorch3d restatement) against scipy."""
import numpy as np
import torch
from scipy.spatial.transform import Rotation as R

from oracle import rotations as T


def _rand_quat(n, seed):
    g = np.random.default_rng(seed)
    q = g.normal(size=(n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def test_quaternion_to_matrix():
    q = _rand_quat(1000, 0)  # wxyz
    m = T.quaternion_to_matrix(torch.tensor(q)).numpy()
    ref = R.from_quat(q[:, [1, 2, 3, 0]]).as_matrix()
    assert np.abs(m - ref).max() < 1e-12


def test_euler_angles_to_matrix_xyz_is_intrinsic():
    g = np.random.default_rng(1)
    a = g.uniform(-np.pi, np.pi, size=(1000, 3))
    m = T.euler_angles_to_matrix(torch.tensor(a), "XYZ").numpy()
    ref = R.from_euler("XYZ", a).as_matrix()
    assert np.abs(m - ref).max() < 1e-12


def test_matrix_to_euler_xyz_roundtrip():
    g = np.random.default_rng(2)
    a = g.uniform(-1.5, 1.5, size=(1000, 3))
    m = T.euler_angles_to_matrix(torch.tensor(a), "XYZ")
    back = T.matrix_to_euler_angles_xyz(m).numpy()
    assert np.abs(back - a).max() < 1e-10
    ref = R.from_matrix(m.numpy()).as_euler("XYZ")
    assert np.abs(back - ref).max() < 1e-10


def test_matrix_to_quaternion():
    q = _rand_quat(2000, 3)
    m = torch.tensor(R.from_quat(q[:, [1, 2, 3, 0]]).as_matrix())
    out = T.matrix_to_quaternion(m).numpy()
    # same rotation up to sign
    sign = np.sign((out * q).sum(1, keepdims=True))
    assert np.abs(out * sign - q).max() < 1e-10
    # near-180-degree rotations exercise the non-w branches
    q2 = q.copy(); q2[:, 0] *= 1e-3; q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
    m2 = torch.tensor(R.from_quat(q2[:, [1, 2, 3, 0]]).as_matrix())
    out2 = T.matrix_to_quaternion(m2).numpy()
    sign = np.sign((out2 * q2).sum(1, keepdims=True))
    assert np.abs(out2 * sign - q2).max() < 1e-9
