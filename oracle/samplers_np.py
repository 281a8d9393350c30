"""CPU restatement of the samplers' transformations (TEST INFRASTRUCTURE; LAB/utils/math.py:1250-1421): what each
sampler does to a given draw u in [0, 1) / z ~ N(0, 1).  Pinned by tests/golden/samplers.npz, which the reference
functions themselves produced from recorded draws (tests/golden/make_golden.py: gen_samplers)."""
import numpy as np

F = np.float32


def sample_uniform(u, lower, upper):
    """:1331  torch.rand(*size) * (upper - lower) + lower"""
    return (F(1) * u * (np.asarray(upper, F) - np.asarray(lower, F)) + np.asarray(lower, F)).astype(F)


def sample_log_uniform(u, lower, upper):
    """:1357-1362"""
    ll, lh = np.log(np.asarray(lower, F)), np.log(np.asarray(upper, F))
    return np.exp((u * (lh - ll) + ll).astype(F)).astype(F)


def sample_triangle(u, lower, upper):
    """:1304-1310"""
    r = (F(2) * u - F(1)).astype(F)
    r = np.where(r < 0, -np.sqrt(np.abs(r)), np.sqrt(np.abs(r))).astype(F)
    r = ((r + F(1)) / F(2)).astype(F)
    return (F(upper - lower) * r + F(lower)).astype(F)


def sample_gaussian(z, mean, std):
    """:1379-1384  torch.normal(mean, std) = mean + std * z"""
    return (np.asarray(mean, F) + np.asarray(std, F) * z).astype(F)


def sample_cylinder(u1, u2, radius, h_range):
    """:1408-1419"""
    ang = ((u1 * F(2) - F(1)) * F(np.pi)).astype(F)
    return np.stack([F(radius) * np.cos(ang), F(radius) * np.sin(ang),
                     (F(h_range[0]) + F(h_range[1] - h_range[0]) * u2)], axis=-1).astype(F)


def random_yaw_orientation(u):
    """:1281-1285 + quat_from_euler_xyz (:251-278) at roll = pitch = 0"""
    yaw = (F(2) * F(np.pi) * u).astype(F)
    half = (yaw * F(0.5)).astype(F)
    z = np.zeros_like(half)
    return np.stack([np.cos(half), z, z, np.sin(half)], axis=-1).astype(F)


def random_orientation(z4):
    """:1265-1267"""
    n = np.sqrt((z4.astype(F) ** 2).sum(-1, keepdims=True)).astype(F)
    return (z4 / np.maximum(n, F(1e-12))).astype(F)
