"""Analytic Planning scene data: the 100 'thin' obstacle variants as capped cylinders.

airgym_amd/assets/thin_trees.json holds radius, length, origin xyz and rpy of the collision cylinder of each
reference asset env_assets/thin/tree_<k>.urdf (extracted by tools/extract_thin_assets.py).  The kernels want,
per variant and in the obstacle's own frame: centre (3), unit axis (3), radius, half length.
"""
import json
import os

import numpy as np

ASSET_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "assets",
                          "thin_trees.json")

NUM_OBSTACLES = 40
CAM_RESOLUTION = (212, 120)     # (W, H): full_camera_array is [N, 1, W, H] (customized.py:144,402)
CAM_CHANNEL = 1


def load_variant_table(path=ASSET_FILE):
    raw = np.asarray(json.load(open(path))["variants"], dtype=np.float64)
    radius, length, ox, oy, oz, roll, pitch, yaw = raw.T
    if np.abs(roll).max() > 1e-9:
        raise ValueError("obstacle variants with a roll component are not supported")
    # URDF rpy = Rz(yaw) Ry(pitch) Rx(roll); the cylinder axis is the local z axis
    axis = np.stack((np.cos(yaw) * np.sin(pitch), np.sin(yaw) * np.sin(pitch), np.cos(pitch)), -1)
    table = np.concatenate((np.stack((ox, oy, oz), -1), axis, radius[:, None], 0.5 * length[:, None]), -1)
    return np.ascontiguousarray(table, dtype=np.float32)
