"""Seeded synthetic Gaussian clouds in the reference's float SoA layout.

Distributions follow SURVEY.md §8(d): positions U(-10,10) (inside the +-2048 range of
the 24-bit fixed point), log-scales U(-8,0), rotations N(0,1)^4 (never zero-norm),
alphas N(0,3), colours N(0,1), sh N(0,0.25).  Draw order: positions, scales,
rotations, alphas, colours, sh.  Array layout is `GaussianCloud`'s
(/root/reference/src/cc/splat-types.h:90-115): flat float32, xyz / xyz / xyzw / a /
rgb / [point][coeff][rgb].
"""
import numpy as np

SH_DIM = {0: 0, 1: 3, 2: 8, 3: 15}
FIELDS = ("positions", "scales", "rotations", "alphas", "colors", "sh")
FLOATS_PER_POINT = {"positions": 3, "scales": 3, "rotations": 4, "alphas": 1, "colors": 3}


def floats_per_point(field, sh_degree):
    if field == "sh":
        return SH_DIM[sh_degree] * 3
    return FLOATS_PER_POINT[field]


def make_cloud_numpy(n, sh_degree, seed):
    """Host cloud as a dict of flat float32 numpy arrays (numpy default_rng(seed))."""
    rng = np.random.default_rng(seed)
    d = SH_DIM[sh_degree] * 3
    cloud = {
        "positions": rng.uniform(-10.0, 10.0, n * 3).astype(np.float32),
        "scales": rng.uniform(-8.0, 0.0, n * 3).astype(np.float32),
        "rotations": rng.standard_normal(n * 4).astype(np.float32),
        "alphas": (rng.standard_normal(n) * 3.0).astype(np.float32),
        "colors": rng.standard_normal(n * 3).astype(np.float32),
        "sh": (rng.standard_normal(n * d) * 0.25).astype(np.float32),
    }
    return cloud


def make_cloud_torch(n, sh_degree, seed, device):
    """Device-resident cloud (dict of flat float32 torch tensors) drawn with a seeded
    torch.Generator on `device`; same distributions as make_cloud_numpy (different
    stream of random numbers)."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    d = SH_DIM[sh_degree] * 3

    def uni(m, lo, hi):
        return torch.empty(m, dtype=torch.float32, device=device).uniform_(lo, hi, generator=g)

    def nrm(m, std):
        return torch.empty(m, dtype=torch.float32, device=device).normal_(0.0, std, generator=g)

    return {
        "positions": uni(n * 3, -10.0, 10.0),
        "scales": uni(n * 3, -8.0, 0.0),
        "rotations": nrm(n * 4, 1.0),
        "alphas": nrm(n, 3.0),
        "colors": nrm(n * 3, 1.0),
        "sh": nrm(n * d, 0.25),
    }
