"""Restatement of the four `pytorch3d.transforms` functions the reference calls
(oracle; test infrastructure).  pytorch3d is an un-vendored, unpinned
dependency (`/root/reference/configuration.sh:89`, `setup.py:24`); its published
algorithm (pytorch3d/transforms/rotation_conversions.py) is restated here and
cross-checked against scipy in tests/test_oracle_rotations.py.

Call sites in the reference: `airgym/envs/base/hovering.py:323-324,338,401-403`,
`airgym/envs/task/tracking.py:173-174,203,244-246`.
All quaternions here are real-first (w, x, y, z) like pytorch3d.
"""
import torch


def quaternion_to_matrix(quaternions: torch.Tensor) -> torch.Tensor:
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k),
            two_s * (i * j - k * r),
            two_s * (i * k + j * r),
            two_s * (i * j + k * r),
            1 - two_s * (i * i + k * k),
            two_s * (j * k - i * r),
            two_s * (i * k - j * r),
            two_s * (j * k + i * r),
            1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(quaternions.shape[:-1] + (3, 3))


def _axis_angle_rotation(axis: str, angle: torch.Tensor) -> torch.Tensor:
    cos = torch.cos(angle)
    sin = torch.sin(angle)
    one = torch.ones_like(angle)
    zero = torch.zeros_like(angle)
    if axis == "X":
        R_flat = (one, zero, zero, zero, cos, -sin, zero, sin, cos)
    elif axis == "Y":
        R_flat = (cos, zero, sin, zero, one, zero, -sin, zero, cos)
    elif axis == "Z":
        R_flat = (cos, -sin, zero, sin, cos, zero, zero, zero, one)
    else:
        raise ValueError("letter must be either X, Y or Z.")
    return torch.stack(R_flat, -1).reshape(angle.shape + (3, 3))


def euler_angles_to_matrix(euler_angles: torch.Tensor, convention: str) -> torch.Tensor:
    """R = R_c0(a0) @ R_c1(a1) @ R_c2(a2)  (intrinsic rotations)."""
    mats = [_axis_angle_rotation(c, e) for c, e in zip(convention, torch.unbind(euler_angles, -1))]
    return torch.matmul(torch.matmul(mats[0], mats[1]), mats[2])


def matrix_to_euler_angles_xyz(matrix: torch.Tensor) -> torch.Tensor:
    """'XYZ' convention only (the only one the reference uses):
    (atan2(-R12, R22), asin(R02), atan2(-R01, R00))."""
    a0 = torch.atan2(-matrix[..., 1, 2], matrix[..., 2, 2])
    a1 = torch.asin(matrix[..., 0, 2])
    a2 = torch.atan2(-matrix[..., 0, 1], matrix[..., 0, 0])
    return torch.stack((a0, a1, a2), -1)


def _sqrt_positive_part(x: torch.Tensor) -> torch.Tensor:
    ret = torch.zeros_like(x)
    positive_mask = x > 0
    ret[positive_mask] = torch.sqrt(x[positive_mask])
    return ret


def matrix_to_quaternion(matrix: torch.Tensor) -> torch.Tensor:
    """Largest-component branch selection (pytorch3d >= 0.5), output (w,x,y,z).
    Sign is not standardised here; the reference re-canonicalises w >= 0 at
    `hovering.py:224-226` before the quaternion is used."""
    batch_dim = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(batch_dim + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(
        torch.stack(
            [
                1.0 + m00 + m11 + m22,
                1.0 + m00 - m11 - m22,
                1.0 - m00 + m11 - m22,
                1.0 - m00 - m11 + m22,
            ],
            dim=-1,
        )
    )
    quat_by_rijk = torch.stack(
        [
            torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
            torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
            torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
        ],
        dim=-2,
    )
    flr = torch.tensor(0.1).to(dtype=q_abs.dtype)
    quat_candidates = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    idx = q_abs.argmax(dim=-1)
    out = torch.gather(quat_candidates, -2, idx[..., None, None].expand(batch_dim + (1, 4))).squeeze(-2)
    return out
