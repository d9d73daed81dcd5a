"""Synthetic ADI cubes for benchmarks, tests and the golden fixtures (SURVEY.md 8(d) generator): stellar halo +
30 speckle modes with a geometric spectrum + unit Gaussian noise, scaled to max|cube| ~ 10, optional planet rotating
with the parallactic angle; angles = linspace(0, 90, n).

``synth_adi`` is THE generator (the CPU checker under tests re-exports it), so the cube ``bench.py`` times is, for the same
(n, N, seed), the cube that tests/golden/g15 (C2), g16 (C3) and g17 (C4) pin against the real reference (float64 random
stream, cast to float32 at the end).  ``synth_adi_device`` builds the same model directly in HBM with torch's generator
for cubes that are too large to draw on the host (C5: 2000 x 1024 x 1024 = 8.4 GB); it is a different random stream."""
import numpy as np


def synth_adi(n, N, seed=0, planet=True, dtype=np.float32):
    """Halo + 30 geometric-spectrum speckle modes + unit noise, max|cube| ~ 10 (float64 stream -> ``dtype``)."""
    rng = np.random.default_rng(seed)
    c = N // 2
    yy, xx = np.mgrid[:N, :N]
    r = np.sqrt((yy - c) ** 2 + (xx - c) ** 2)
    env = np.exp(-r / (N / 8))
    nmodes = 30
    modes = rng.standard_normal((nmodes, N, N)) * env
    coef = rng.standard_normal((n, nmodes)) * 2.0 ** (-np.arange(nmodes) / 3)
    cube = np.tensordot(coef, modes, axes=1) + env[None] * 3.0
    angles = np.linspace(0, 90, n)
    if planet:
        sig = 4 / 2.3548200450309493
        for i, th in enumerate(np.deg2rad(angles)):
            py, px = c + (N / 4) * np.sin(th), c + (N / 4) * np.cos(th)
            cube[i] += 0.5 * np.exp(-((yy - py) ** 2 + (xx - px) ** 2) / (2 * sig ** 2))
    cube *= 9.0 / np.max(np.abs(cube))
    cube += rng.standard_normal((n, N, N))
    cube *= 10.0 / np.max(np.abs(cube))
    return cube.astype(dtype), angles


def synth_adi_device(n, N, seed=0, device=None, chunk=100, nmodes=30, halving=3.0):
    """The model of ``synth_adi`` (without the planet) drawn on the GPU: returns (float32 cuda tensor (n, N, N) with
    max|cube| ~ 10, float64 numpy angles).  Deterministic for a given seed on a given device type.

    ``nmodes`` speckle modes whose amplitudes halve every ``halving`` modes (defaults = the SURVEY 8(d) generator: 30 modes,
    of which ~10 stand above the unit noise, so that a boundary at ncomp = 20 .. 50 lies INSIDE the flat noise bulk -- the
    worst case for every leading-subspace method).  More modes with a slower decay (e.g. 150, 10) give the decaying spectrum
    of real quasi-static speckle data, with the boundary on the slope."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
    g = torch.Generator(device=dev).manual_seed(int(seed))
    yy, xx = torch.meshgrid(torch.arange(N, device=dev), torch.arange(N, device=dev), indexing="ij")
    env = torch.exp(-torch.sqrt((yy - N // 2) ** 2.0 + (xx - N // 2) ** 2.0) / (N / 8)).float()
    nmodes = int(nmodes)
    modes = torch.randn((nmodes, N, N), device=dev, generator=g) * env
    coef = torch.randn((n, nmodes), device=dev, generator=g) * (2.0 ** (-torch.arange(nmodes, device=dev) / float(halving)))
    cube = torch.empty((n, N, N), device=dev)
    peak = 0.0
    for i in range(0, n, chunk):
        m = min(chunk, n - i)
        cube[i:i + m] = torch.tensordot(coef[i:i + m], modes, dims=1) + 3 * env
        peak = max(peak, float(cube[i:i + m].abs().max()))
    s1 = 9.0 / peak
    peak = 0.0
    for i in range(0, n, chunk):
        m = min(chunk, n - i)
        cube[i:i + m] = cube[i:i + m] * s1 + torch.randn((m, N, N), device=dev, generator=g)
        peak = max(peak, float(cube[i:i + m].abs().max()))
    cube *= 10.0 / peak
    return cube, np.linspace(0, 90, n)
