"""GPU parity at BASELINE.json's full C2 size (400 x 512 x 512, ncomp = 20), where the oracle is too slow to run on the
whole cube: size-independent properties of the path (orthonormal PCs, residuals orthogonal to them, recon + residuals
= data, linearity of the derotation, exact quarter turns), plus the oracle on a handful of frames / pixel rows."""
import numpy as np
import pytest

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu

N_FR, N_PX, K = 400, 512, 20


@pytest.fixture(scope="module")
def c2():
    import torch
    from vip_amd.psfsub import pca
    cube, ang = O.synth_adi(N_FR, N_PX, seed=0)        # the generator (and seed) of tests/golden/g15_pca_c2.npz
    cube_t = torch.from_numpy(cube).cuda()
    out = pca(cube_t, ang, ncomp=K, full_output=True, verbose=False, check_memory=False)
    return cube, ang, cube_t, out


def test_c2_projection_properties(c2):
    import torch
    cube, ang, cube_t, (frame, pcs, recon, res, res_der) = c2
    P = N_PX * N_PX
    V = pcs.reshape(K, P).double()
    M = cube_t.reshape(N_FR, P)
    R = res.reshape(N_FR, P)
    assert (V @ V.T - torch.eye(K, dtype=torch.float64, device=V.device)).abs().max().item() < 2e-5   # orthonormal PCs
    # reconstructed + residuals = data (float32 round-off of one subtraction)
    assert (recon.reshape(N_FR, P) + R - M).abs().max().item() < 4e-6
    # residuals are orthogonal to the PCs: |V r| small against |r|
    c = (R.double() @ V.T).abs().max().item()
    assert c < 1e-3 * R.double().norm(dim=1).max().item() / np.sqrt(K)
    # the model lives in the row space of the data: recon = (M V^T) V
    coeff = M.double() @ V.T
    assert ((coeff @ V).float() - recon.reshape(N_FR, P)).abs().max().item() < 1e-4
    # idempotence: projecting the residuals again removes nothing more
    again = R.double() - (R.double() @ V.T) @ V
    assert (again.float() - R).abs().max().item() < 1e-4


def test_c2_derotation_and_median_against_oracle_samples(c2):
    cube, ang, cube_t, (frame, pcs, recon, res, res_der) = c2
    ang_c = O.check_pa_vector(ang)
    for i in (0, 57, 199, 266, 399):                 # derotation angles in several rot90 quadrants
        exp = O.frame_rotate_fft(res[i].cpu().numpy().astype(np.float64), -ang_c[i])
        got = res_der[i].cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        assert np.nanmax(np.abs(got - exp)) < 1e-4, i
    rows = slice(200, 216)
    exp = np.nanmedian(res_der[:, rows].cpu().numpy(), axis=0)
    assert np.array_equal(frame[rows].cpu().numpy(), exp)          # median collapse: bit-exact


def test_c2_derotation_linearity_and_quarter_turns():
    import torch
    from vip_amd import backend as B
    rng = np.random.default_rng(5)
    n = 24
    a = torch.from_numpy(rng.standard_normal((n, N_PX, N_PX)).astype(np.float32)).cuda()
    b = torch.from_numpy(rng.standard_normal((n, N_PX, N_PX)).astype(np.float32)).cuda()
    angles = np.linspace(-170, 190, n)
    da, db, dab = B.derotate(a, angles), B.derotate(b, angles), B.derotate(a + 2 * b, angles)
    assert (da + 2 * db - dab).abs().max().item() < 1e-4
    # quarter turns are exact re-indexings (rot90 about pixel N//2 through the odd embedding): no interpolation error
    for ang, k in ((-90.0, 1), (-180.0, 2), (-270.0, 3), (90.0, 3)):
        got = B.derotate(a[:2], np.array([ang, ang]))[0].cpu().numpy()
        src = a[0].cpu().numpy()
        emb = np.zeros((N_PX + 1, N_PX + 1), np.float32)
        emb[:N_PX, :N_PX] = src
        exp = np.rot90(emb, k)[:N_PX, :N_PX]
        assert np.abs(got - exp).max() < 2e-5, ang


def test_c2_against_the_real_reference(c2):
    """BASELINE.json configs[1] pinned directly: tests/golden/g15_pca_c2.npz holds outputs of the real reference's
    pca(cube, angles, ncomp=20, svd_mode='lapack') on this very cube (oracle/gen_golden_c2.py, ~15 min of CPU)."""
    from conftest import load_golden
    g = load_golden("g15_pca_c2")
    cube, ang, cube_t, (frame, pcs, recon, res, res_der) = c2
    assert np.abs(frame.cpu().numpy() - g["frame"]).max() < 1e-4
    keep = [int(i) for i in g["keep"]]
    assert np.abs(res[keep].cpu().numpy() - g["res_keep"]).max() < 1e-4
    assert np.nanmax(np.abs(res_der[keep].cpu().numpy() - g["resd_keep"])) < 1e-4
    rows = res.reshape(N_FR, -1).double().sum(dim=1).cpu().numpy()
    assert np.abs(rows - g["res_rowsum"]).max() < 1e-4 * np.sqrt(N_PX * N_PX) * 4


def test_c3_annular_against_the_real_reference():
    """BASELINE.json configs[2]: pca_annular(400 x 512 x 512, 8 annuli, ncomp=10) against the real reference's run
    (oracle/gen_golden_c3c4.py c3 -> tests/golden/g16_annular_c3.npz)."""
    import torch
    from conftest import load_golden
    import os
    from conftest import ROOT
    from vip_amd.psfsub import pca_annular
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", "g16_annular_c3.npz")):
        pytest.skip("g16_annular_c3.npz not generated (the reference run takes more than an hour of CPU)")
    g = load_golden("g16_annular_c3")
    cube, ang = O.synth_adi(N_FR, N_PX, seed=0)
    co, cd, fr = pca_annular(torch.from_numpy(cube).cuda(), ang, ncomp=10, asize=32, fwhm=4, delta_rot=(0.1, 1),
                             n_segments=1, full_output=True, verbose=False)
    assert np.abs(fr.cpu().numpy() - g["frame"]).max() < 1e-4
    keep = [int(i) for i in g["keep"]]
    assert np.abs(co[keep].cpu().numpy() - g["out_keep"]).max() < 1e-4
    rows = co.reshape(N_FR, -1).double().sum(dim=1).cpu().numpy()
    assert np.abs(rows - g["out_rowsum"]).max() < 1e-4 * N_PX * 4


def test_c4_per_channel_pca_against_the_real_reference():
    """BASELINE.json configs[3]: pca(39 x 200 x 256 x 256, ncomp=20) per channel + spectral mean against the real
    reference's run (oracle/gen_golden_c3c4.py c4 -> tests/golden/g17_pca4d_c4.npz); both the batched frame-only path
    and the per-channel loop of full_output."""
    import torch
    from conftest import load_golden
    from vip_amd.psfsub import pca
    g = load_golden("g17_pca4d_c4")
    c4 = torch.stack([torch.from_numpy(O.synth_adi(200, 256, seed=s)[0]) for s in range(39)]).cuda()
    ang = np.linspace(0, 90, 200)
    frame = pca(c4, ang, ncomp=20, verbose=False, check_memory=False)
    assert np.abs(frame.cpu().numpy() - g["frame"]).max() < 1e-4
    out = pca(c4, ang, ncomp=20, full_output=True, verbose=False, check_memory=False)
    chs = [int(c) for c in g["ifs_channels"]]
    assert np.abs(out[-1][chs].cpu().numpy() - g["ifs"]).max() < 1e-4
    assert np.abs(out[0].cpu().numpy() - g["frame"]).max() < 1e-4
