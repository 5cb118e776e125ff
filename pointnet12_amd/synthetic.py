"""Synthetic point clouds for tests and benchmarks (host-side numpy; no device code).

``kitti_cloud`` is the "KITTI-shaped" distribution of SURVEY.md §8(d): a 64-beam scanner
model seen through the +-40 degree "inview" window (reference data_utils/kitti_utils.py:222),
normalised the way the reference loader does it (data_utils/SemKITTI_Loader.py:23-30:
x/70, y/70, z/3, (i-0.5)*2, clip to [-1,1]) and resampled WITH replacement
(SemKITTI_Loader.py:111), so duplicate points reach FPS exactly as they do in training.
"""
import numpy as np

SEED_BASE = 20260101


def kitti_cloud(seed, n_points, n_raw=None, extra_dims=5):
    """One cloud -> float32 [n_points, 3 + 1 + extra_dims] (xyz, intensity, extra channels)."""
    if n_raw is None:
        n_raw = 20000 if n_points <= 8192 else 120000
    rng = np.random.default_rng(seed)
    beam = rng.integers(0, 64, size=n_raw)
    phi = np.deg2rad(np.linspace(-24.8, 2.0, 64))[beam]
    theta = np.deg2rad(rng.uniform(-40.0, 40.0, size=n_raw))
    with np.errstate(divide="ignore"):
        rho_ground = np.where(phi < np.deg2rad(-1.0), 1.73 / np.tan(-phi), np.inf)
    rho_obst = 5.0 + rng.exponential(20.0, size=n_raw)
    rho = np.minimum(np.minimum(rho_ground, rho_obst), 70.0)
    xyz = np.stack([rho * np.cos(phi) * np.cos(theta),
                    rho * np.cos(phi) * np.sin(theta),
                    rho * np.sin(phi)], axis=1)
    xyz = xyz + rng.normal(0.0, 0.02, size=xyz.shape)
    intensity = rng.uniform(0.0, 1.0, size=(n_raw, 1))
    extra = rng.uniform(-1.0, 1.0, size=(n_raw, extra_dims))
    pcd = np.concatenate([xyz[:, 0:1] / 70.0, xyz[:, 1:2] / 70.0, xyz[:, 2:3] / 3.0,
                          (intensity - 0.5) * 2.0, extra], axis=1)
    pcd = np.clip(pcd, -1.0, 1.0).astype(np.float32)
    choice = rng.choice(n_raw, n_points, replace=True)
    return pcd[choice]


def kitti_batch(first_cloud, batch, n_points, channels=9):
    """Clouds first_cloud .. first_cloud+batch-1 -> (points [B, channels, N] f32, labels [B, N] int64).

    ``channels`` = 3 + feature dims (9 for the S3DIS-style 3+6 input of the benchmark,
    4 for the KITTI xyz+intensity input of the shipped checkpoint).
    """
    assert 3 <= channels <= 9
    pts = np.stack([kitti_cloud(SEED_BASE + first_cloud + i, n_points)[:, :channels] for i in range(batch)])
    labels = np.stack([np.random.default_rng(SEED_BASE + first_cloud + i + 10 ** 6).integers(0, 13, size=n_points)
                       for i in range(batch)]).astype(np.int64)
    return np.ascontiguousarray(pts.transpose(0, 2, 1)), labels


def uniform_batch(seed, batch, n_points, channels=9):
    """Sparse stress case: U(-1,1)^3 positions (about 2.5 neighbours inside r=0.1 at N=4096)."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-1.0, 1.0, size=(batch, channels, n_points)).astype(np.float32)
    labels = rng.integers(0, 13, size=(batch, n_points)).astype(np.int64)
    return pts, labels
