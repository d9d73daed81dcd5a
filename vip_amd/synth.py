"""Synthetic ADI cubes for benchmarks / smoke tests (SURVEY.md 8(d) generator): stellar halo +
30 speckle modes with a geometric spectrum + unit Gaussian noise, scaled to max|cube| ~ 10, optional
planet rotating with the parallactic angle.  float32, C order; angles = linspace(0, 90, n)."""
import numpy as np


def synth_adi(n, N, seed=0, planet=True, dtype=np.float32):
    rng = np.random.default_rng(seed)
    c = N // 2
    yy, xx = np.mgrid[:N, :N]
    r = np.sqrt((yy - c) ** 2 + (xx - c) ** 2)
    env = np.exp(-r / (N / 8)).astype(np.float32)
    nmodes = 30
    modes = rng.standard_normal((nmodes, N, N), dtype=np.float32) * env
    coef = (rng.standard_normal((n, nmodes)) * 2.0 ** (-np.arange(nmodes) / 3)).astype(np.float32)
    cube = np.tensordot(coef, modes, axes=1) + env[None] * 3.0
    angles = np.linspace(0, 90, n)
    if planet:
        sig = 4 / 2.3548200450309493
        for i, th in enumerate(np.deg2rad(angles)):
            py, px = c + (N / 4) * np.sin(th), c + (N / 4) * np.cos(th)
            cube[i] += (0.5 * np.exp(-((yy - py) ** 2 + (xx - px) ** 2) / (2 * sig ** 2))).astype(np.float32)
    cube *= np.float32(9.0 / np.max(np.abs(cube)))
    cube += rng.standard_normal((n, N, N), dtype=np.float32)
    cube *= np.float32(10.0 / np.max(np.abs(cube)))
    return cube.astype(dtype, copy=False), angles
