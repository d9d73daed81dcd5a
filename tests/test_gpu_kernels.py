"""GPU parity tests, kernel level: every entry point of the C ABI (through vip_amd.backend) against the
CPU oracle / numpy float64 on the same seeded inputs, plus the committed golden fixtures."""
import numpy as np
import pytest

from conftest import load_golden, sign_align
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    import torch
    assert torch.cuda.is_available()
    from vip_amd import backend
    return backend


def dev(B, a):
    return B.to_device_f32(a)


@pytest.fixture(autouse=True)
def _exact_eigensolvers(request, B):
    """The test_eigh_topk* tests pin the EXACT tridiagonal kernels (bit-identical vectors with and without the whole
    spectrum, projector accuracy at 1e-10 across gaps of 1e-6): the verified fast path (eigh_chfsi.hip, n >= 700) is
    switched off for them and has its own tests below."""
    exact = request.node.name.startswith("test_eigh_topk")
    fast = "fast" in request.node.name
    ctx = B.get_context()
    if exact:
        ctx.set_option("eigh_fast", 0)
    if fast:
        ctx.set_option("eigh_fast_min", 256)      # (by default the fast path only runs where it pays: from 700 .. 1000 rows)
    yield
    if exact:
        ctx.set_option("eigh_fast", 1)
    if fast:
        ctx.set_option("eigh_fast_min", 0)


# ---- Gram -------------------------------------------------------------------------------------

@pytest.mark.parametrize("n,P", [(5, 77), (16, 256), (50, 16384), (33, 10201), (100, 4096), (400, 8192)])
@pytest.mark.parametrize("f32acc", [0, 1])
def test_gram(B, n, P, f32acc):
    rng = np.random.default_rng(n * 7 + P)
    M = (rng.standard_normal((n, P)) * 3 + 0.5).astype(np.float32)
    # asymmetric content so that a transposed tile write would be caught
    M[:, : min(P, 64)] += np.arange(n, dtype=np.float32)[:, None]
    ctx = B.get_context()
    ctx.set_option("gram_f32", f32acc)
    try:
        G = B.gram(dev(B, M)).cpu().numpy()
    finally:
        ctx.set_option("gram_f32", 0)
    ref = M.astype(np.float64) @ M.astype(np.float64).T
    scale = np.abs(ref).max()
    tol = 3e-6 if f32acc else 1e-12
    assert np.abs(G - ref).max() <= tol * scale
    assert np.array_equal(G, G.T)


def test_gram_tile_variants(B):
    rng = np.random.default_rng(5)
    M = rng.standard_normal((70, 2048)).astype(np.float32)
    ref = M.astype(np.float64) @ M.astype(np.float64).T
    ctx = B.get_context()
    for tb in (1, 2, 3, 4):
        ctx.set_option("gram_tb", tb)
        G = B.gram(dev(B, M)).cpu().numpy()
        assert np.abs(G - ref).max() <= 1e-12 * np.abs(ref).max(), tb
    ctx.set_option("gram_tb", 0)


@pytest.mark.parametrize("batch,n,P", [(7, 39, 1089), (3, 12, 4096), (5, 30, 333), (2, 70, 2500)])
def test_gram_batched(B, batch, n, P):
    """One launch for a stack of equally shaped Gram problems (ADI+mSDI first pass); odd row lengths included."""
    rng = np.random.default_rng(batch * 100 + n)
    M = (rng.standard_normal((batch, n, P)) * 2 + 0.3).astype(np.float32)
    G = B.gram_batched(dev(B, M)).cpu().numpy()
    ref = np.einsum("bip,bjp->bij", M.astype(np.float64), M.astype(np.float64))
    assert G.shape == (batch, n, n)
    assert np.abs(G - ref).max() < 1e-9 * np.abs(ref).max()
    assert np.array_equal(G, G.transpose(0, 2, 1))


def test_cross_gram(B):
    rng = np.random.default_rng(6)
    A = rng.standard_normal((37, 3000)).astype(np.float32)
    Bm = rng.standard_normal((9, 3000)).astype(np.float32)
    C = B.cross_gram(dev(B, A), dev(B, Bm)).cpu().numpy()
    ref = A.astype(np.float64) @ Bm.astype(np.float64).T
    assert C.shape == (37, 9)
    assert np.abs(C - ref).max() <= 1e-12 * np.abs(ref).max()


# ---- eigensolver -------------------------------------------------------------------------------

@pytest.mark.parametrize("n", [3, 16, 17, 50, 64, 100, 200, 400, 650])
def test_eigh(B, n):
    import torch
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, 3 * n + 5))
    M[:, 0] *= 30
    M[:, 1] *= 10
    G = M @ M.T
    evals, evecs = B.eigh(torch.from_numpy(G).cuda())
    evals, evecs = evals.cpu().numpy(), evecs.cpu().numpy()
    w, v = np.linalg.eigh(G)
    w, v = w[::-1], v[:, ::-1].T
    assert np.all(np.diff(evals) <= 0)
    np.testing.assert_allclose(evals, w, rtol=1e-10, atol=1e-10 * w[0])
    # orthonormal rows, and G v = lambda v
    assert np.abs(evecs @ evecs.T - np.eye(n)).max() < 1e-9
    assert np.abs(G @ evecs.T - evecs.T * evals).max() < 1e-9 * w[0]
    k = min(n, 10)
    assert np.abs(sign_align(evecs[:k], v[:k]) - v[:k]).max() < 1e-7


def test_eigh_batched_and_rank_deficient(B):
    import torch
    rng = np.random.default_rng(0)
    Gs = []
    for i in range(5):
        M = rng.standard_normal((40, 30 + 3 * i))      # rank < n for the first ones
        Gs.append(M @ M.T)
    Gs = np.stack(Gs)
    evals, evecs = B.eigh(torch.from_numpy(Gs).cuda())
    evals, evecs = evals.cpu().numpy(), evecs.cpu().numpy()
    for i in range(5):
        w = np.linalg.eigvalsh(Gs[i])[::-1]
        np.testing.assert_allclose(evals[i], w, atol=1e-10 * w[0])
        r = 30 + 3 * i if 30 + 3 * i < 40 else 40
        E = evecs[i, :r]
        assert np.abs(E @ E.T - np.eye(r)).max() < 1e-8


def _topk_check(G, ev, ec, k, tol_proj=1e-10):
    n = G.shape[0]
    w, v = np.linalg.eigh(G)
    w, v = w[::-1], v[:, ::-1]
    scale = max(abs(w[0]), 1e-300)
    np.testing.assert_allclose(ev, w[:k], atol=1e-12 * scale)
    assert np.all(np.diff(ev) <= 1e-12 * scale)
    X = ec.T                                               # n x k
    assert np.abs(X.T @ X - np.eye(k)).max() < 1e-11
    assert np.abs(G @ X - X * ev).max() < 1e-11 * scale
    # sign convention: the largest-magnitude component of every vector is positive
    assert np.all(X[np.abs(X).argmax(axis=0), np.arange(k)] > 0)
    # projector onto the leading invariant subspace (well defined when lambda_k > lambda_k+1)
    if k < n and (w[k - 1] - w[k]) > 1e-6 * scale:
        assert np.abs(X @ X.T - v[:, :k] @ v[:, :k].T).max() < tol_proj


@pytest.mark.parametrize("n,k", [(1, 1), (2, 1), (2, 2), (3, 2), (17, 5), (64, 64), (65, 10), (128, 20), (200, 10),
                                 (400, 20), (512, 64)])
def test_eigh_topk(B, n, k):
    import torch
    rng = np.random.default_rng(n * 100 + k)
    M = rng.standard_normal((n, 3 * n + 5))
    M[:, 0] *= 30
    M[:, 1] *= 10
    G = M @ M.T
    ev, ec = B.eigh_topk(torch.from_numpy(G.copy()).cuda(), k)
    _topk_check(G, ev.cpu().numpy(), ec.cpu().numpy(), k)


@pytest.mark.parametrize("n,k", [(150, 100), (300, 128), (512, 65), (700, 100), (1100, 200)])
def test_eigh_topk_more_than_64_vectors(B, n, k):
    """k > 64: the matrix-in-L2 solver loops over the vectors of a workgroup (the one-sided Jacobi kernel it replaces
    here did not converge on graded spectra for n >~ 600).  Graded spectrum as a PSF-dominated Gram matrix has."""
    import torch
    rng = np.random.default_rng(n + k)
    M = rng.standard_normal((n, 2 * n + 7)) * np.logspace(0, -3, 2 * n + 7)
    G = M @ M.T
    ev, ec = B.eigh_topk(torch.from_numpy(G.copy()).cuda(), k)
    _topk_check(G, ev.cpu().numpy(), ec.cpu().numpy(), k)
    ev2, ec2 = B.eigh_topk(torch.from_numpy(G.copy()).cuda(), k, all_evals=True)
    assert np.array_equal(ec2.cpu().numpy(), ec.cpu().numpy())
    w = np.linalg.eigvalsh(G)[::-1]
    assert np.abs(ev2.cpu().numpy() - w).max() < 1e-11 * w[0]


@pytest.mark.parametrize("n,k", [(2049, 5), (2500, 30), (2200, 150), (3100, 20)])
def test_eigh_topk_more_than_2048_rows(B, n, k):
    """2048 < n <= 6144: three vectors of n doubles in LDS, gathered vectors in registers, factors of the inverse
    iteration in global memory (eigh_tri_large.hip: tri_xl_kernel); k = 150 > 128 workgroups loops over the vectors."""
    import torch
    rng = np.random.default_rng(n + k)
    M = rng.standard_normal((n, n + 50)) * np.logspace(0, -2, n + 50)
    M[:, :4] *= 20
    G = M @ M.T
    ev, ec = B.eigh_topk(torch.from_numpy(G.copy()).cuda(), k)
    _topk_check(G, ev.cpu().numpy(), ec.cpu().numpy(), k)
    if n <= 2500:
        ev2, ec2 = B.eigh_topk(torch.from_numpy(G.copy()).cuda(), k, all_evals=True)
        assert np.array_equal(ec2.cpu().numpy(), ec.cpu().numpy())
        w = np.linalg.eigvalsh(G)[::-1]
        assert np.abs(ev2.cpu().numpy() - w).max() < 1e-11 * w[0]


def test_eigh_topk_degenerate_and_padded(B):
    import torch
    rng = np.random.default_rng(5)
    n, k = 48, 6
    mats = []
    M = rng.standard_normal((n, 200)); M[1] = M[0]; M[7] = M[0]                  # duplicated frames: exact null space
    mats.append(M @ M.T)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.r_[5.0, 5.0, 5.0, 3.0, 3.0, 1.0, np.zeros(n - 6)]                    # exactly repeated leading eigenvalues
    mats.append((Q * lam) @ Q.T)
    mats.append(np.zeros((n, n)))                                                 # zero matrix
    mats.append(np.diag(np.arange(n, 0, -1.0)))                                   # already diagonal (all reflectors trivial)
    T = np.diag(rng.random(n) + 1) + np.diag(rng.random(n - 1), 1); T = T + np.triu(T, 1).T
    mats.append(T)                                                                # already tridiagonal
    G = np.stack(mats)
    ev, ec = B.eigh_topk(torch.from_numpy(G.copy()).cuda(), k)
    ev, ec = ev.cpu().numpy(), ec.cpu().numpy()
    for i in (0, 3, 4):
        _topk_check(G[i], ev[i], ec[i], k)
    # repeated eigenvalues: the eigenvalues, orthonormality and the invariant subspace of the 6 non-zero ones
    np.testing.assert_allclose(ev[1], lam[:k], atol=1e-12)
    X = ec[1].T
    assert np.abs(X.T @ X - np.eye(k)).max() < 1e-11
    assert np.abs(X @ X.T - Q[:, :k] @ Q[:, :k].T).max() < 1e-9
    assert np.abs(ev[2]).max() == 0.0 and np.all(np.isfinite(ec[2]))
    # zero padding with per-problem active sizes (annular PCA: libraries of different length)
    sizes = np.array([48, 30, 7, 1, 2], dtype=np.int32)
    Gp = np.zeros_like(G)
    for i, m in enumerate(sizes):
        A = rng.standard_normal((m, 60))
        Gp[i, :m, :m] = A @ A.T
    ev, ec = B.eigh_topk(torch.from_numpy(Gp.copy()).cuda(), k, nact=torch.from_numpy(sizes).cuda())
    ev, ec = ev.cpu().numpy(), ec.cpu().numpy()
    for i, m in enumerate(sizes):
        kk = min(k, m)
        _topk_check(Gp[i, :m, :m], ev[i, :kk], ec[i, :kk, :m], kk)
        assert np.all(ec[i, :, m:] == 0) and np.all(ev[i, kk:] == 0) and np.all(ec[i, kk:] == 0)


def test_eigh_topk_matches_jacobi_in_pca(B):
    """The fused PCA gives the same residuals with either eigensolver (option eigh_method)."""
    from vip_amd.synth import synth_adi
    cube, _ = synth_adi(60, 64, seed=3)
    M = dev(B, cube.reshape(60, -1))
    ctx = B.get_context()
    ctx.set_option("eigh_method", 1)
    r1 = B.pca_project(M, 7)[0].cpu().numpy()
    ctx.set_option("eigh_method", 0)
    r2 = B.pca_project(M, 7)[0].cpu().numpy()
    assert np.abs(r1 - r2).max() < 2e-6


# ---- projection kernels ---------------------------------------------------------------------------

@pytest.mark.parametrize("n,k,P", [(12, 3, 1024), (50, 5, 16384), (37, 20, 10201), (100, 33, 4096), (64, 64, 640)])
def test_rowspace_and_subtract_gemm(B, n, k, P):
    rng = np.random.default_rng(n + k)
    M = rng.standard_normal((n, P)).astype(np.float32)
    W = rng.standard_normal((k, n)).astype(np.float32)
    rs = (rng.random(k) + 0.5).astype(np.float32)
    ctx = B.get_context()
    Md, Wd, rsd = dev(B, M), dev(B, W), dev(B, rs)
    T = B.empty((k, P))
    ctx.call("vipmi_rowspace_gemm_f32", B.ptr(Wd), B.ptr(Md), k, n, P, B.ptr(rsd), B.ptr(T))
    refT = (W.astype(np.float64) @ M.astype(np.float64)) * rs[:, None]
    assert np.abs(T.cpu().numpy() - refT).max() <= 2e-5 * np.abs(refT).max()
    C = rng.standard_normal((n, k)).astype(np.float32)
    Tm = rng.standard_normal((k, P)).astype(np.float32)
    R = B.empty((n, P))
    recon = B.empty((n, P))
    Cd, Tmd = dev(B, C), dev(B, Tm)          # keep the device copies alive across the call
    ctx.call("vipmi_subtract_gemm_f32", B.ptr(Md), B.ptr(Cd), B.ptr(Tmd), n, k, P, B.ptr(R), B.ptr(recon))
    refrec = C.astype(np.float64) @ Tm.astype(np.float64)
    assert np.abs(recon.cpu().numpy() - refrec).max() <= 2e-5 * np.abs(refrec).max()
    assert np.abs(R.cpu().numpy() - (M - refrec)).max() <= 2e-5 * np.abs(refrec).max()
    R2 = B.empty((n, P))
    ctx.call("vipmi_subtract_gemm_f32", B.ptr(Md), B.ptr(Cd), B.ptr(Tmd), n, k, P, B.ptr(R2), B.ptr(None))
    assert np.array_equal(R.cpu().numpy(), R2.cpu().numpy())


@pytest.mark.parametrize("n,P,k", [(40, 1000, 3), (100, 16384, 20), (33, 515, 32), (70, 1000, 33), (120, 4099, 50), (200, 2048, 64), (150, 777, 100),
                                   (140, 1536, 128), (140, 1536, 129)])
def test_subtract_lds_tile_is_bit_identical_to_the_register_kernel(B, n, P, k):
    """The subtraction stages the tile of T in LDS, shared by the four waves of a workgroup, which split the frame blocks
    (same accumulation order: bit-identical to the register kernel, option subtract_lds = 0; more than 128 components: the
    register kernel); residuals + reconstruction = input rows, residuals orthogonal to the PCs."""
    import torch
    rng = np.random.default_rng(n + k)
    M = torch.from_numpy(rng.standard_normal((n, P)).astype(np.float32)).cuda()
    ctx = B.get_context()
    out = {}
    try:
        for mode in (0, 1):
            ctx.set_option("subtract_lds", mode)
            res, recon, pcs, _ = B.pca_project(M, k, want_recon=True, want_pcs=True)
            out[mode] = (res.clone(), recon.clone(), pcs.clone())
            res_only = B.pca_project(M, k)[0]
            assert torch.equal(res_only, res)
    finally:
        ctx.set_option("subtract_lds", 1)
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
    res, recon, pcs = out[1]
    assert float((res + recon - M).abs().max()) < 1e-5
    assert float((res.double() @ pcs.double().T).abs().max()) < 2e-3 * float(M.abs().max()) * np.sqrt(P) / 30


def test_pca_project_matches_golden(B):
    g = load_golden("g2_project_subtract")
    cube = g["cube"]
    M = dev(B, cube.reshape(cube.shape[0], -1))
    res, recon, pcs, evals = B.pca_project(M, 3, want_recon=True, want_pcs=True, want_evals=True)
    assert np.abs(res.cpu().numpy().reshape(cube.shape) - g["res_None_None"]).max() < 1e-4
    V = pcs.cpu().numpy()
    assert np.abs(V @ V.T - np.eye(3)).max() < 1e-5
    g1 = load_golden("g1_svd")
    for tag in ("a", "b"):
        Md = dev(B, g1["M_" + tag])
        _, _, pcs, evals = B.pca_project(Md, 6, want_pcs=True, want_evals=True)
        Vr = g1["V_lapack_" + tag]
        assert np.abs(sign_align(pcs.cpu().numpy(), Vr) - Vr).max() < 3e-5
        np.testing.assert_allclose(np.sqrt(evals.cpu().numpy()[:6]), g1["S_lapack_" + tag], rtol=1e-5)


# ---- scaling / mask --------------------------------------------------------------------------------

@pytest.mark.parametrize("mode", ["temp-mean", "temp-standard", "spat-mean", "spat-standard"])
def test_scale(B, mode):
    rng = np.random.default_rng(3)
    m = (rng.standard_normal((12, 999)) * 3 + 5).astype(np.float32)
    m[:, 7] = 2.5
    out = B.scale(dev(B, m), mode).cpu().numpy()
    ref = O.matrix_scaling(m, mode)
    assert np.abs(out - ref).max() < 5e-6
    out2 = B.scale(dev(B, m), mode, out=None)
    assert np.array_equal(out, out2.cpu().numpy())


def test_mask(B):
    import torch
    from vip_amd.var import mask_circle
    g = load_golden("g5_indices")
    out = mask_circle(g["mask_in"], 5)
    assert np.array_equal(out, g["mask_out_5"])
    rng = np.random.default_rng(1)
    a2 = rng.standard_normal((33, 33)).astype(np.float32)
    assert np.array_equal(mask_circle(a2, 6.5), O.mask_circle(a2, 6.5))
    assert np.array_equal(mask_circle(a2, 0), a2)


# ---- collapse -----------------------------------------------------------------------------------------

@pytest.mark.parametrize("n", [7, 8])
def test_collapse_golden(B, n):
    from vip_amd.preproc import cube_collapse
    g = load_golden("g4_collapse")
    for mode in ("median", "max"):
        for tag in ("", "_nan"):
            got = cube_collapse(g["cube%s_%d" % (tag, n)], mode)
            exp = g["%s%s_%d" % (mode, tag, n)]
            assert got.dtype == np.float32
            assert np.array_equal(got, exp, equal_nan=True), (mode, tag)
    for mode in ("mean", "sum", "absmean"):
        for tag in ("", "_nan"):
            got = cube_collapse(g["cube%s_%d" % (tag, n)], mode)
            exp = g["%s%s_%d" % (mode, tag, n)]
            assert np.array_equal(np.isnan(got), np.isnan(exp))
            assert np.nanmax(np.abs(got - exp)) < 2e-6
    got = cube_collapse(g["cube_%d" % n], "wmean", w=g["w_%d" % n])
    assert np.abs(got - g["wmean_%d" % n]).max() < 2e-6


@pytest.mark.parametrize("n,P", [(1, 100), (2, 100), (63, 1000), (64, 1000), (65, 333), (129, 257), (400, 4096), (1000, 70),
                                 (1025, 130), (1500, 70), (2048, 33), (2000, 4096),       # (1025 .. 2048: keys in registers, chunked staging)
                                 (4097, 300), (6001, 130), (5000, 64)])      # (more than 4096 frames: the streaming kernel)
def test_median_sizes_bitexact(B, n, P):
    rng = np.random.default_rng(n + P)
    cube = rng.standard_normal((n, P)).astype(np.float32)
    cube[rng.random((n, P)) < 0.02] = np.nan
    cube[:, 3] = np.nan
    cube[:, 5] = 1.25                                  # all duplicates
    cube[: n // 2, 6] = -0.0
    cube[n // 2:, 6] = 0.0
    got = B.collapse(dev(B, cube).reshape(n, P, 1), "median").cpu().numpy().reshape(P)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = np.nanmedian(cube, axis=0)
    assert np.array_equal(got, exp, equal_nan=True)


@pytest.mark.parametrize("n", [5, 64, 100, 399, 400, 448, 1000, 2000, 4096])
def test_median_bucket_selection_hard_cases(B, n):
    """The median kernel bins the samples of a pixel between their minimum and maximum and finishes exactly on the bin
    that holds the wanted rank: columns built to defeat the binning (outliers that squeeze everything else into one bin,
    heavy ties, two clusters with the median at the gap, +-0, infinities, a range that overflows float32, denormals)."""
    rng = np.random.default_rng(n)
    P = 64
    cube = rng.standard_normal((n, P)).astype(np.float32)
    cube[0, 0] = 1e30                                   # one outlier: every other sample lands in bin 0 -> second level
    cube[0, 1], cube[1 % n, 1] = -3e38, 3e38            # fhi - flo overflows -> bisection
    cube[:, 2] = np.round(cube[:, 2] * 2) / 2           # heavy ties (9 distinct values)
    cube[:, 3] = np.where(np.arange(n) % 2 == 0, -5.0, 5.0) + 1e-3 * cube[:, 3]    # two clusters, median at the gap
    cube[:, 4] = np.where(np.arange(n) % 3 == 0, -0.0, 0.0)
    cube[0, 5], cube[n - 1, 5] = -np.inf, np.inf
    cube[:, 6] = (cube[:, 6] * 1e-42).astype(np.float32)                           # denormals
    cube[:, 7] = 7.0
    cube[0, 7] = 1e20                                   # all ties but one outlier
    cube[:, 8] = np.float32(1.0) + np.arange(n, dtype=np.float32) * np.float32(1.1920929e-07)   # consecutive floats
    cube[1:, 9] = np.nan                                # one valid sample
    cube[:, 10] = np.exp(8 * cube[:, 10])               # log-normal: long tail, crowded first bins
    cube[:, 11] = np.where(rng.random(n) < 0.7, 0.0, cube[:, 11])                  # 70 % exact zeros (masked frames)
    cube[rng.random((n, P)) < 0.05] = np.nan
    cube[:, 12] = np.nan
    got = B.collapse(dev(B, cube).reshape(n, P, 1), "median").cpu().numpy().reshape(P)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = np.nanmedian(cube, axis=0)
    assert np.array_equal(got, exp, equal_nan=True), np.nonzero(~((got == exp) | (np.isnan(got) & np.isnan(exp))))


@pytest.mark.parametrize("n,P,tn", [(7, 50, 3), (8, 50, 3), (8, 64, 4), (7, 33, 9), (7, 33, 50), (65, 333, 20),
                                    (400, 1024, 100), (129, 100, 129), (10, 40, 1), (1500, 40, 700), (2048, 64, 50),
                                    (4100, 70, 50), (5001, 40, 1001)])      # (more than 4096 frames: the streaming kernel)
def test_trimmean_sizes(B, n, P, tn):
    rng = np.random.default_rng(n * 7 + P + tn)
    cube = rng.standard_normal((n, P)).astype(np.float32)
    cube[rng.random((n, P)) < 0.05] = np.nan
    cube[:, 3] = np.nan
    cube[:, 5] = 1.25
    cube[: n // 2, 6] = -2.0
    cube[n // 2:, 6] = 3.0
    cube[:, 7] = np.round(cube[:, 7])                  # many duplicates at the slice edges
    got = B.collapse(dev(B, cube).reshape(n, P, 1), "trimmean", trim_n=tn).cpu().numpy().reshape(P)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = O.cube_collapse(cube.reshape(n, P, 1), "trimmean", n=tn).reshape(P)
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    assert np.nanmax(np.abs(got - exp)) < 2e-6


def test_trimmean_golden(B):
    from vip_amd.preproc import cube_collapse
    g = load_golden("g4_collapse")
    for n in (7, 8):
        got = cube_collapse(g["cube_%d" % n], "trimmean", n=3)
        assert np.abs(got - g["trimmean_%d" % n]).max() < 2e-6


# ---- rotation (generic direct path) ------------------------------------------------------------------

@pytest.mark.parametrize("N", [32, 33, 64])
def test_frame_rotate_golden_direct(B, N):
    from vip_amd.preproc import frame_rotate
    g = load_golden("g3_rotate")
    fr = g["frame_%d" % N]
    for i, th in enumerate(g["angles"]):
        got = frame_rotate(fr, th, method="direct")
        assert got.dtype == np.float64
        assert np.abs(got - g["rot_%d" % N][i]).max() < 2e-5, th


def test_rotate_masks_and_cube_direct(B):
    from vip_amd.preproc import frame_rotate, cube_derotate
    g = load_golden("g3_rotate")
    got = frame_rotate(g["frame_nan"], 33.0, method="direct")
    assert np.array_equal(np.isnan(got), np.isnan(g["rot_nan"]))
    assert np.nanmax(np.abs(got - g["rot_nan"])) < 2e-5
    got = frame_rotate(g["frame_zero"], 33.0, mask_val=0, interp_zeros=True, ker=1, method="direct")
    assert np.abs(got - g["rot_zero"]).max() < 2e-5
    got = cube_derotate(g["cube_33"], g["cube_33_angles"], method="direct")
    assert got.dtype == np.float32
    assert np.abs(got - g["derot_33"]).max() < 5e-5
    got = cube_derotate(g["cube_128"], g["cube_128_angles"], method="direct")
    assert np.abs(got - g["derot_128"]).max() < 5e-5


@pytest.mark.parametrize("N", [513, 577, 640, 801, 1001, 1023])
def test_derotate_513_to_1023_px_vs_oracle(B, N):
    """513 .. 1023 px: the convolution passes take every line in two parts (derotate_conv.inc) -- against the float64
    restatement of the reference and against the direct correlations; all four rot90 quadrants, NaN mask."""
    from vip_amd.preproc import cube_derotate
    rng = np.random.default_rng(N)
    angles = np.array([10.0, 100.0, 200.0, 300.0])
    cube = rng.standard_normal((4, N, N)).astype(np.float32)
    cube[1, 3:6, 4] = np.nan
    cube[2, N - 5, N - 7:] = np.nan
    ref = O.cube_derotate(cube, angles)
    ctx = B.get_context()
    try:
        got = cube_derotate(cube, angles, method="direct")
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        assert np.nanmax(np.abs(got - ref)) < 4e-5
        if N in (513, 801):
            ctx.set_option("rot_conv", 0)
            slow = cube_derotate(cube, angles, method="direct")
            assert np.nanmax(np.abs(slow - got)) < 4e-5
    finally:
        ctx.set_option("rot_conv", 1)


@pytest.mark.parametrize("N", [1025, 1100, 1537])
def test_derotate_beyond_1024_px(B, N):
    """1025 .. 2048 px: three or four parts per line on the 1024-point convolutions, against the direct correlations
    (and the float64 restatement at 1100 px)"""
    from vip_amd.preproc import cube_derotate
    rng = np.random.default_rng(N)
    angles = np.array([33.0, 250.0])
    cube = rng.standard_normal((2, N, N)).astype(np.float32)
    cube[1, 3:6, 4] = np.nan
    ctx = B.get_context()
    try:
        got = cube_derotate(cube, angles, method="direct")
        ctx.set_option("rot_conv", 0)
        slow = cube_derotate(cube, angles, method="direct")
        assert np.array_equal(np.isnan(got), np.isnan(slow))
        assert np.nanmax(np.abs(slow - got)) < 6e-5
        if N == 1100:
            ref = O.cube_derotate(cube, angles)
            assert np.nanmax(np.abs(got - ref)) < 6e-5
    finally:
        ctx.set_option("rot_conv", 1)


@pytest.mark.parametrize("N", [21, 101, 129, 200, 255, 256, 301, 400, 511, 512])
def test_derotate_generic_sizes_vs_oracle(B, N):
    """Non-power-of-two padded lengths take the real-split direct path (from 129 px its passes run as power-of-two
    circular convolutions, derotate_conv.inc) -- every rot90 quadrant, angles whose shears land next to integer shifts,
    NaN mask."""
    from vip_amd.preproc import cube_derotate
    rng = np.random.default_rng(N)
    angles = np.array([10.0, 50.0, 100.0, 200.0, 300.0, 359.0, -44.9, 45.1, 180.0, 0.0])
    n = len(angles) if N < 300 else 5
    cube = rng.standard_normal((n, N, N)).astype(np.float32)
    cube[1, 3:6, 4] = np.nan
    ref = O.cube_derotate(cube, angles[:n])
    ctx = B.get_context()
    try:
        got = cube_derotate(cube, angles[:n], method="direct")
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        assert np.nanmax(np.abs(got - ref)) < 2e-5
        if 128 < N <= 512:                      # the convolution passes against the direct correlations they replace
            ctx.set_option("rot_conv", 0)
            slow = cube_derotate(cube, angles[:n], method="direct")
            assert np.nanmax(np.abs(slow - ref)) < 2e-5
    finally:
        ctx.set_option("rot_conv", 1)


@pytest.mark.parametrize("N", [80, 81])
def test_cube_derotate_reference_roundtrip(B, N):
    # reference tests/pre_3_10/test_preproc_rotation.py:21-69: 24 successive derotations of ones
    from vip_amd.preproc import cube_derotate
    arr = np.ones((4, N, N), np.float32)
    angs = np.array([120, 90, 60, 45.])
    cur = arr.copy()
    for _ in range(24):
        cur = cube_derotate(cur, angs)
    c0 = N // 2 - 25
    assert np.allclose(cur[:, c0:c0 + 50, c0:c0 + 50], 1.0, rtol=1e-1, atol=1e-1)


# ---- rotation: FFT path (power-of-two padded lengths) ------------------------------------------------

@pytest.mark.parametrize("N", [128, 256, 512])
def test_derotate_fft_vs_oracle_and_direct(B, N):
    from vip_amd.preproc import cube_derotate
    rng = np.random.default_rng(N)
    angles = np.array([3.0, -47.5, 95.0, 200.1, 333.3, 135.0, 44.999, 270.0])
    n = len(angles) if N < 512 else 4
    cube = rng.standard_normal((n, N, N)).astype(np.float32)
    cube[0, 5:9, 7] = np.nan
    got = cube_derotate(cube, angles[:n], method="fft")
    ref = O.cube_derotate(cube, angles[:n])
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.nanmax(np.abs(got - ref)) < 2e-5
    if N <= 256:
        alt = cube_derotate(cube, angles[:n], method="direct")
        assert np.nanmax(np.abs(got - alt)) < 2e-5


@pytest.mark.parametrize("N", [128, 256])
def test_derotate_batching_is_bit_identical(B, N):
    """The derotation works through the cube in batches sized by the workspace option; results must not depend on
    the batch size (deterministic kernels, blocked intermediates indexed by the frame's position in its batch)."""
    import torch
    rng = np.random.default_rng(N)
    ang = np.array([3.0, -47.5, 95.0, 200.1, 333.3, 135.0, 44.999, 270.0, 181.0, 91.0, -1.0])
    cube = torch.from_numpy(rng.standard_normal((len(ang), N, N)).astype(np.float32)).cuda()
    ctx = B.get_context()
    ref = B.derotate(cube, ang).cpu().numpy()
    try:
        for rb in (1, 3, 4):
            ctx.set_option("rot_batch", rb)
            assert np.array_equal(B.derotate(cube, ang).cpu().numpy(), ref), rb
    finally:
        ctx.set_option("rot_batch", 0)


@pytest.mark.parametrize("N", [128, 256, 512])
def test_derotate_paired_store_layout_is_bit_identical(B, N):
    """rot_pair_store: another layout of the intermediate between the column shear and the last row shear (32-byte stores);
    the arithmetic is the same, so are the bits (N = 128 has an odd number of row groups: the option is ignored there)."""
    import torch
    rng = np.random.default_rng(N + 1)
    ang = np.array([3.0, -47.5, 95.0, 200.1, 333.3, 135.0, 44.999, 270.0, 181.0])
    cube = rng.standard_normal((len(ang), N, N)).astype(np.float32)
    cube[2, :5, :7] = np.nan
    ct = torch.from_numpy(cube).cuda()
    ctx = B.get_context()
    ref = B.derotate(ct, ang).cpu().numpy()
    try:
        ctx.set_option("rot_pair_store", 1)
        got = B.derotate(ct, ang).cpu().numpy()
    finally:
        ctx.set_option("rot_pair_store", 0)
    assert np.array_equal(got, ref, equal_nan=True)


def test_derotate_fft_edge_cases(B):
    """Few frames (fewer tasks than workgroups in the dynamic queues), a single frame, an all-NaN frame, a constant
    frame and angles that are exact multiples of 90 / 360 degrees."""
    from vip_amd.preproc import cube_derotate
    rng = np.random.default_rng(7)
    for n in (1, 2, 3):
        cube = rng.standard_normal((n, 128, 128)).astype(np.float32)
        ang = np.array([33.3, -120.0, 361.0])[:n]
        got = cube_derotate(cube, ang, method="fft")
        assert np.nanmax(np.abs(got - O.cube_derotate(cube, ang))) < 2e-5, n
    cube = rng.standard_normal((4, 128, 128)).astype(np.float32)
    cube[1] = np.nan
    cube[2] = 3.5
    ang = np.array([0.0, 77.0, 180.0, -360.0])
    got = cube_derotate(cube, ang, method="fft")
    ref = O.cube_derotate(cube, ang)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.isnan(got[1]).all()
    assert np.nanmax(np.abs(got - ref)) < 2e-5
    assert np.abs(got[0] - cube[0]).max() < 5e-6                          # angle 0: identity up to FFT round-off


def test_derotate_fft_golden_128(B):
    from vip_amd.preproc import cube_derotate
    g = load_golden("g3_rotate")
    got = cube_derotate(g["cube_128"], g["cube_128_angles"], method="fft")
    assert np.abs(got - g["derot_128"]).max() < 5e-5
    got0 = cube_derotate(g["cube_128"], g["cube_128_angles"], mask_val=0, method="fft")
    assert np.abs(got0 - g["derot_128"]).max() < 5e-5      # no exact zeros in this cube


@pytest.mark.parametrize("w1", [1, 0])
def test_derotate_fft_1024_vs_oracle(B, w1):
    """Le = 4096, both plans (one wave per line and per SIMD with blocked intermediates / two cooperating waves with the
    tiled column shear): the oracle on three frames in different rot90 quadrants."""
    from vip_amd.preproc import cube_derotate
    rng = np.random.default_rng(1024)
    angles = np.array([17.3, 100.0, 250.5])
    cube = rng.standard_normal((3, 1024, 1024)).astype(np.float32)
    cube[1, 7:9, 11] = np.nan
    ctx = B.get_context()
    ctx.set_option("rot_4096_w1", w1)
    try:
        got = cube_derotate(cube, angles, method="fft")
    finally:
        ctx.set_option("rot_4096_w1", 1)
    ref = O.cube_derotate(cube, angles)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.nanmax(np.abs(got - ref)) < 3e-5


def test_derotate_fft_1024_delta(B):
    """Le = 4096 plan: delta-function known answers and a round trip."""
    from vip_amd.preproc import cube_derotate
    N = 1024
    c = N // 2
    d = np.zeros((3, N, N), np.float32)
    d[:, c, c + 10] = 1
    out = cube_derotate(d, np.array([-90.0, -180.0, -30.0]), method="fft")     # frame_rotate by +90, +180, +30
    for i, (dy, dx, val) in enumerate(((-10, 0, 1.0), (0, -10, 1.0), (-5, 9, 0.8207))):
        iy, ix = np.unravel_index(np.argmax(out[i]), out[i].shape)
        assert (iy - c, ix - c) == (dy, dx)
        assert abs(out[i][iy, ix] - val) < 6e-4
    rng = np.random.default_rng(0)
    sm = rng.standard_normal((1, N, N)).astype(np.float32)
    ref = O.cube_derotate(sm, np.array([12.5]))
    got = cube_derotate(sm, np.array([12.5]), method="fft")
    assert np.abs(got - ref).max() < 2e-5


@pytest.mark.parametrize("N", [128, 256, 512])
def test_derotate_fft_vs_oracle_all_quadrants(B, N):
    """real-split two-for-one transforms (pruned butterflies, canonical positions) vs the float64 oracle."""
    from vip_amd.preproc import cube_derotate
    rng = np.random.default_rng(N + 1)
    angles = np.array([7.0, -47.5, 95.0, 200.1, 333.3, 135.0])
    n = len(angles) if N < 512 else 3
    cube = (rng.standard_normal((n, N, N)) * 3).astype(np.float32)
    got = cube_derotate(cube, angles[:n], method="fft")
    ref = O.cube_derotate(cube, angles[:n])
    assert np.abs(got - ref).max() < 5e-5


def test_eigh_topk_repeatable_under_load():
    """The multi-workgroup solver synchronises through counters in global memory: many back-to-back launches of
    different sizes (8, 16 and 32 cooperating workgroups) must all give the numpy answer, bit-identically per size."""
    import torch
    rng = np.random.default_rng(99)
    mats = {}
    for n in (128, 200, 256, 400, 512):
        M = rng.standard_normal((n, 2 * n))
        mats[n] = M @ M.T
    first = {}
    for rep in range(6):
        for n, G in mats.items():
            ev, ec = B_eigh(G, 12)
            w = np.linalg.eigvalsh(G)[::-1][:12]
            np.testing.assert_allclose(ev, w, atol=1e-12 * w[0])
            if n in first:
                assert np.array_equal(first[n][0], ev) and np.array_equal(first[n][1], ec)
            else:
                first[n] = (ev, ec)


def test_eigh_topk_one_xcd_exchange_matches_the_spread_layout(B):
    """eigh_one_xcd: the cooperating workgroups of a problem on one XCD, exchanging through its L2 (placement verified
    with HW_REG_XCC_ID in the kernel) -- bit-identical to the agent-scope exchange of the spread layout, for one problem and
    for a batch (problems on different XCDs), and under concurrent launches from several threads (the XCD rotates per launch)."""
    import threading
    import torch
    rng = np.random.default_rng(5)
    ctx = B.get_context()
    mats = {}
    for n in (130, 256, 300, 400, 448, 512):
        M = rng.standard_normal((n, 2 * n))
        mats[n] = M @ M.T
    try:
        res = {}
        ctx.set_option("eigh_wave", 0)        # the LDS-resident kernel in both layouts (the register-resident one: next test)
        for mode in (0, 1):
            ctx.set_option("eigh_one_xcd", mode)
            for n, G in mats.items():
                res[mode, n] = B_eigh(G, 15)
            Gb = torch.from_numpy(np.stack([mats[300] * (1.0 + 0.1 * i) for i in range(5)])).cuda()
            ev, ec = B.eigh_topk(Gb, 9)
            res[mode, "batch"] = (ev.cpu().numpy(), ec.cpu().numpy())
        for key in [k for k in res if k[0] == 0]:
            a, b = res[key], res[1, key[1]]
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), key
        w = np.linalg.eigvalsh(mats[400])[::-1][:15]
        np.testing.assert_allclose(res[1, 400][0], w, atol=1e-12 * w[0])
        # four threads, each on its own stream / context, all in the one-XCD mode at the same time
        out, errs = {}, []

        def work(i):
            try:
                with torch.cuda.stream(torch.cuda.Stream()):
                    c = B.get_context()
                    c.set_option("eigh_one_xcd", 1)
                    c.set_option("eigh_wave", 0)
                    c.set_option("eigh_fast", 0)
                    for rep in range(8):
                        ev_, ec_ = B.eigh_topk(torch.from_numpy(mats[400].copy()).cuda(), 15)
                        out[i, rep] = (ev_.cpu().numpy(), ec_.cpu().numpy())
                    c.set_option("eigh_one_xcd", -1)
                    c.set_option("eigh_wave", 1)
                    c.set_option("eigh_fast", 1)
            except Exception as e:           # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        for v in out.values():
            assert np.array_equal(v[0], res[1, 400][0]) and np.array_equal(v[1], res[1, 400][1])
    finally:
        ctx.set_option("eigh_one_xcd", -1)
        ctx.set_option("eigh_wave", 1)


def test_eigh_topk_wave_resident_tridiagonalisation(B):
    """A lone synchronous problem of 129 .. 448 rows: Householder reduction on 64 cooperating single-wave workgroups with the
    matrix in registers (eigh_wave.hip), stages 2-5 of the LDS kernel as a second launch.  Against numpy float64 (eigenvalues,
    residuals, the invariant subspace), against the LDS-resident reduction, deterministic from call to call, and from four
    threads at once (each launch on its own XCD)."""
    import threading
    import torch
    rng = np.random.default_rng(11)
    ctx = B.get_context()
    try:
        ref400 = None
        for n, k in ((129, 3), (192, 64), (200, 10), (257, 7), (320, 40), (385, 5), (400, 20), (448, 20)):
            M = rng.standard_normal((n, n + 40)) * (2.0 ** (-np.arange(n + 40) / 60.0))
            G = M @ M.T
            ctx.set_option("eigh_wave", 1)
            ev, ec = B_eigh(G, k)
            ev2, ec2 = B_eigh(G, k)
            assert np.array_equal(ev, ev2) and np.array_equal(ec, ec2), n
            _topk_check(G, ev, ec, k)
            ctx.set_option("eigh_wave", 0)
            ev0, ec0 = B_eigh(G, k)
            np.testing.assert_allclose(ev, ev0, atol=1e-12 * ev0[0])
            assert np.abs(ec.T @ ec - ec0.T @ ec0).max() < 1e-8, n      # same invariant subspace
            if n == 400:
                ref400 = (G, ev, ec)
        ctx.set_option("eigh_wave", 1)
        G, ev, ec = ref400
        out, errs = {}, []

        def work(i):
            try:
                with torch.cuda.stream(torch.cuda.Stream()):
                    c = B.get_context()
                    c.set_option("eigh_fast", 0)
                    for rep in range(8):
                        ev_, ec_ = B.eigh_topk(torch.from_numpy(G.copy()).cuda(), 20)
                        out[i, rep] = (ev_.cpu().numpy(), ec_.cpu().numpy())
                    c.set_option("eigh_fast", 1)
                    B.check_deferred()
            except Exception as e:           # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        for v in out.values():
            assert np.array_equal(v[0], ev) and np.array_equal(v[1], ec)
    finally:
        ctx.set_option("eigh_wave", 1)


def _recovered(ctx):
    return max(0, ctx.get_option("eigh_recovered"))


def test_eigh_wave_time_out_recovers(B):
    """The 64 waves of the wave-resident reduction spin on each other; when one of them never becomes resident (here: the test
    hook eigh_wave_drop launches one short; in the field: a chip-filling kernel of ANOTHER process holds its CU) every wave runs
    into the time-out (~1 s) and leaves.  The recovery launch that follows every cooperating solve then solves the problem again
    on the single-workgroup kernel: right eigenpairs, no error, a counter (`eigh_recovered`).  With eigh_recover = 0 the
    round-4 behaviour: NaN eigenpairs, vipmi_check_deferred reports the time-out.  The next call is unaffected either way."""
    import torch
    from vip_amd import backend
    rng = np.random.default_rng(3)
    M = rng.standard_normal((300, 700))
    G = M @ M.T
    ctx = B.get_context()
    assert ctx.lib.vipmi_check_deferred(ctx.handle) == 0
    good = B_eigh(G, 8)
    rec0 = _recovered(ctx)
    try:
        ctx.set_option("eigh_wave_drop", 1)
        ev, ec = B_eigh(G, 8)
        _topk_check(G, ev, ec, 8)
        np.testing.assert_allclose(ev, good[0], rtol=1e-11)
        assert ctx.lib.vipmi_check_deferred(ctx.handle) == 0
        assert _recovered(ctx) == rec0 + 1
        # whole spectrum + leading vectors: stages 2-5 run as a second cooperating launch, which must not consume the void reduction
        evs, ecs = backend.eigh_topk(torch.from_numpy(G.copy()).cuda(), 8, all_evals=True)
        evs, ecs = evs.cpu().numpy(), ecs.cpu().numpy()
        w = np.linalg.eigvalsh(G)[::-1]
        np.testing.assert_allclose(evs, w, atol=1e-12 * w[0])
        _topk_check(G, evs[:8], ecs, 8)
        assert ctx.lib.vipmi_check_deferred(ctx.handle) == 0
        assert _recovered(ctx) == rec0 + 2
        ctx.set_option("eigh_recover", 0)
        ev, ec = B_eigh(G, 8)
        assert np.isnan(ev).all() and np.isnan(ec).all()
        assert ctx.lib.vipmi_check_deferred(ctx.handle) != 0
    finally:
        ctx.set_option("eigh_wave_drop", 0)
        ctx.set_option("eigh_recover", 1)
    assert ctx.lib.vipmi_check_deferred(ctx.handle) == 0
    again = B_eigh(G, 8)
    assert np.array_equal(good[0], again[0]) and np.array_equal(good[1], again[1])
    assert _recovered(ctx) == rec0 + 2


@pytest.mark.parametrize("batch,n,k", [(1, 400, 20), (3, 300, 12), (2, 120, 5)])
def test_eigh_multi_time_out_recovers(B, batch, n, k):
    """The same for the multi-workgroup kernel (8-32 workgroups per problem behind counter barriers; the pipelined mode's solver and
    the one for a few problems at a time): a participant that never arrives (hook eigh_multi_drop) -> every barrier gives up, the
    recovery launch solves all problems of the launch again."""
    import torch
    from vip_amd import backend
    rng = np.random.default_rng(100 * batch + n)
    Gs = []
    for _ in range(batch):
        M = rng.standard_normal((n, n + 150)) * (2.0 ** (-np.arange(n + 150) / 60.0))
        Gs.append(M @ M.T)
    Gs = np.stack(Gs)
    ctx = B.get_context()
    assert ctx.lib.vipmi_check_deferred(ctx.handle) == 0
    rec0 = _recovered(ctx)
    try:
        ctx.set_option("eigh_wave", 0)          # a lone problem also takes the multi-workgroup kernel
        ctx.set_option("eigh_reg", 0)           # (n <= 200: not the single-workgroup register kernel)
        ctx.set_option("eigh_multi_drop", 1)
        ev, ec = backend.eigh_topk(torch.from_numpy(Gs.copy()).cuda(), k)
        ev, ec = ev.cpu().numpy(), ec.cpu().numpy()
        for b in range(batch):
            _topk_check(Gs[b], ev[b], ec[b], k)
        assert ctx.lib.vipmi_check_deferred(ctx.handle) == 0
        assert _recovered(ctx) == rec0 + 1
    finally:
        ctx.set_option("eigh_multi_drop", 0)
        ctx.set_option("eigh_wave", 1)
        ctx.set_option("eigh_reg", 1)
    ev2, ec2 = backend.eigh_topk(torch.from_numpy(Gs.copy()).cuda(), k)
    for b in range(batch):
        _topk_check(Gs[b], ev2[b].cpu().numpy(), ec2[b].cpu().numpy(), k)
    assert ctx.lib.vipmi_check_deferred(ctx.handle) == 0 and _recovered(ctx) == rec0 + 1


def B_eigh(G, k):
    import torch
    from vip_amd import backend
    ev, ec = backend.eigh_topk(torch.from_numpy(G.copy()).cuda(), k)
    return ev.cpu().numpy(), ec.cpu().numpy()


@pytest.mark.parametrize("n,k", [(513, 5), (640, 64), (641, 7), (700, 30), (1000, 50), (1536, 64)])
def test_eigh_topk_large(B, n, k):
    """640 < n <= 2048: the matrix stays in global memory, 64 cooperating workgroups (eigh_tri_large.hip; 513 .. 640 rows:
    the LDS-resident kernel on 32 workgroups); graded
    spectrum (dynamic range 2^-n/20), on which a Jacobi sweep count would explode."""
    import torch
    rng = np.random.default_rng(n + k)
    M = rng.standard_normal((n, n + 50)) * (2.0 ** (-np.arange(n + 50) / 40.0))
    G = M @ M.T
    ev, ec = B.eigh_topk(torch.from_numpy(G.copy()).cuda(), k)
    _topk_check(G, ev.cpu().numpy(), ec.cpu().numpy(), k)


@pytest.mark.parametrize("n,k", [(40, 3), (200, 8), (400, 20), (700, 5)])
def test_eigh_spectrum(B, n, k):
    """All eigenvalues + leading k vectors (what CEVR / svd_wrapper(full_output) need) from the tridiagonal solvers:
    single-workgroup, LDS-resident multi-workgroup and global-memory variants."""
    import torch
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n + 20)) * (2.0 ** (-np.arange(n + 20) / 30.0))
    G = M @ M.T
    ev, ec = B.eigh_topk(torch.from_numpy(G.copy()).cuda(), k, all_evals=True)
    ev, ec = ev.cpu().numpy(), ec.cpu().numpy()
    w = np.linalg.eigvalsh(G)[::-1]
    assert ev.shape == (n,) and ec.shape == (k, n)
    np.testing.assert_allclose(ev, w, atol=1e-12 * w[0])
    _topk_check(G, ev[:k], ec, k)


@pytest.mark.parametrize("N", [7, 64, 65, 101])
@pytest.mark.parametrize("interp", ["nearneig", "bilinear", "bicubic", "lanczos4"])
def test_rotate_opencv_style(N, interp):
    """imlib='opencv' (derotation.py:279-305): the warp kernel against the oracle's restatement of OpenCV's warpAffine
    (fixed-point coordinates, 1/32-pixel weight tables, zero border; cv2 itself is absent => parity unpinned), plus
    the exact cases (0 / 90 degrees) and closeness to the 'vip-fft' rotation on a smooth frame."""
    from vip_amd.preproc import cube_derotate, frame_rotate
    rng = np.random.default_rng(N)
    yy, xx = np.mgrid[:N, :N]
    smooth = np.exp(-((yy - 0.6 * N) ** 2 + (xx - 0.4 * N) ** 2) / (0.02 * N * N)).astype(np.float32)
    cube = np.stack([smooth, rng.standard_normal((N, N)).astype(np.float32), smooth, smooth, smooth])
    cube[1, 3, 5] = np.nan
    ang = np.array([33.0, -147.3, 0.0, -90.0, 211.7])
    got = cube_derotate(cube, ang, imlib="opencv", interpolation=interp)
    assert got.dtype == np.float32 and got.shape == cube.shape
    for i in range(len(ang)):
        ref = O.warp_rotate(cube[i], -ang[i], interp)
        assert np.abs(got[i] - ref).max() < 2e-6 * max(1.0, np.abs(ref).max()), (i, interp)
    assert np.array_equal(got[2], smooth)
    assert np.array_equal(got[3], np.rot90(smooth, 1)[:N, :N]) or N % 2 == 0
    fft = cube_derotate(cube[:1], ang[:1])
    b = N // 6
    assert N < 32 or np.abs(got[0] - fft[0])[b:-b, b:-b].max() < {"nearneig": 0.2, "bilinear": 0.05, "bicubic": 0.03, "lanczos4": 0.02}[interp]
    one = frame_rotate(cube[0], 12.5, imlib="opencv", interpolation=interp, cxy=(N / 2 - 0.5, N / 2 - 0.5))
    assert one.dtype == np.float32
    assert np.abs(one - O.warp_rotate(cube[0], 12.5, interp, cxy=(N / 2 - 0.5, N / 2 - 0.5))).max() < 2e-6


def test_rotate_opencv_style_errors():
    from vip_amd.preproc import cube_derotate
    cube = np.zeros((2, 16, 16), dtype=np.float32)
    with pytest.raises(ValueError):
        cube_derotate(cube, [0, 1], imlib="opencv", interpolation="spline")
    with pytest.raises(ValueError):
        cube_derotate(cube, [0, 1], imlib="opencv", border_mode="nope")
    with pytest.raises(NotImplementedError):
        cube_derotate(cube, [0, 1], imlib="skimage")
    with pytest.raises(ValueError):
        cube_derotate(cube, [0, 1], imlib="nope")


@pytest.mark.parametrize("mode", ["median", "mean", "sum", "max", "absmean", "wmean", "trimmean"])
def test_collapse_batched_equals_per_cube(mode):
    """one launch over a stack of cubes (blockIdx.y = cube) gives exactly the per-cube collapses, NaNs included"""
    import torch
    from vip_amd import backend as B
    rng = np.random.default_rng(5)
    stack = rng.standard_normal((5, 39, 23, 23)).astype(np.float32)
    stack[1, 3, 4, 5] = np.nan
    stack[2, :, 0, 0] = np.nan
    t = torch.from_numpy(stack).cuda()
    w = rng.random(39).astype(np.float32) if mode == "wmean" else None
    kw = dict(trim_n=11) if mode == "trimmean" else {}
    got = B.collapse_batched(t, mode, w=w, **kw).cpu().numpy()
    exp = np.stack([B.collapse(t[b], mode, w=w, **kw).cpu().numpy() for b in range(5)])
    assert got.shape == (5, 23, 23) and np.array_equal(got, exp, equal_nan=True)
    if mode == "median":
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                assert np.array_equal(got, np.nanmedian(stack, axis=1), equal_nan=True)


@pytest.mark.parametrize("shape", [(7, 39, 3, 33 * 33), (3, 20, 5, 64 * 64), (2, 70, 40, 1000)])
def test_project_batched_equals_per_problem(shape):
    """vipmi_project_batched_f32 = the rowspace + subtract products of every problem (same kernels, blockIdx = problem)"""
    import torch
    from vip_amd import backend as B
    nb, n, k, P = shape
    rng = np.random.default_rng(nb * n)
    M = torch.from_numpy(rng.standard_normal((nb, n, P)).astype(np.float32)).cuda()
    Q = np.stack([np.linalg.qr(rng.standard_normal((n, k)))[0].T for _ in range(nb)]).astype(np.float32)
    E = torch.from_numpy(Q).cuda().contiguous()
    got = B.project_batched(M, E).cpu().numpy()
    ctx = B.get_context(0)
    for b in range(nb):
        T = B.empty((k, P), device=0)
        R = B.empty((n, P), device=0)
        Cb = E[b].t().contiguous()
        ctx.call("vipmi_rowspace_gemm_f32", B.ptr(E[b]), B.ptr(M[b]), k, n, P, None, B.ptr(T))
        ctx.call("vipmi_subtract_gemm_f32", B.ptr(M[b]), B.ptr(Cb), B.ptr(T), n, k, P, B.ptr(R), None)
        assert np.array_equal(got[b], R.cpu().numpy()), b
    Mn = M.cpu().numpy().astype(np.float64)
    ref = Mn - np.einsum("bkn,bkp->bnp", Q.astype(np.float64), np.einsum("bkn,bnp->bkp", Q.astype(np.float64), Mn))
    assert np.abs(got - ref).max() < 2e-5


@pytest.mark.parametrize("border", ["edge", "symmetric", "reflect", "wrap"])
@pytest.mark.parametrize("interp", ["nearneig", "bilinear", "lanczos4"])
def test_rotate_opencv_style_borders(border, interp):
    """the reference's border_mode switch of imlib='opencv' (derotation.py:294-305 -> cv2.BORDER_REPLICATE / REFLECT /
    REFLECT_101 / WRAP): kernel against the oracle (cv::borderInterpolate restated; np.pad pins the index maps)"""
    from vip_amd.preproc import cube_derotate
    rng = np.random.default_rng(3)
    for N in (9, 50):
        cube = rng.standard_normal((3, N, N)).astype(np.float32)
        ang = np.array([17.0, -63.0, 200.5])
        got = cube_derotate(cube, ang, imlib="opencv", interpolation=interp, border_mode=border)
        for i in range(3):
            ref = O.warp_rotate(cube[i], -ang[i], interp, border_mode=border)
            assert np.abs(got[i] - ref).max() < 5e-6 * max(1.0, np.abs(ref).max()), (N, i)
        const = cube_derotate(cube, ang, imlib="opencv", interpolation=interp)
        assert not np.array_equal(got, const)


# ---- verified fast path of the leading-k eigensolver (eigh_chfsi.hip) -----------------------------------------------

def _spectrum_matrix(n, lam, seed=0):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    G = (Q * lam) @ Q.T
    return 0.5 * (G + G.T)


def _baseline_like(n, seed=0):
    """A few geometric modes above a narrow noise bulk: the spectrum of the BASELINE generator."""
    rng = np.random.default_rng(seed)
    lam = 0.024 * (1.0 + 0.08 * np.sort(rng.uniform(-1, 1, n))[::-1] ** 3 + 0.06 * np.linspace(1, -1, n))
    lam[:10] += 2.0 ** (-1.2 * np.arange(10))
    return np.sort(lam)[::-1]


@pytest.mark.parametrize("n,k,kind", [(700, 20, "bulk"), (1000, 30, "bulk"), (2000, 50, "bulk"), (900, 50, "bulk"), (777, 17, "decay"),
                                      (640, 12, "repeated")])
def test_eigh_fast_path_is_verified_and_deterministic(B, n, k, kind):
    """Chebyshev-filtered subspace iteration: every returned pair has ||G q - theta q|| <= 1e-12 theta_1 (gate 1e-13 at the
    Ritz level), eigenvalues and orthogonality at round-off, the leading subspace to residual / gap, the sign convention of
    the exact solvers; two runs are bit-identical (fixed start block, fixed reduction orders)."""
    import torch
    lam = {"bulk": _baseline_like(n, seed=n), "decay": (1.0 + np.arange(n)) ** -1.5,
           "repeated": np.r_[np.repeat([5.0, 3.0, 2.0], 4), 0.5 * 0.97 ** np.arange(n - 12)]}[kind]
    G = _spectrum_matrix(n, lam, seed=n + k)
    ctx = B.get_context()
    ctx.set_option("eigh_fast", 1)
    Gt = torch.from_numpy(G).cuda()
    ev, ec = B.eigh_topk(Gt.clone(), k)
    info = [ctx.get_option("eigh_fast_last_" + s) for s in ("products", "rounds", "locked", "reason")]
    assert info[3] == 0 and info[2] == k, info                       # converged on the fast path, not by fall-back
    assert torch.equal(Gt, torch.from_numpy(G).cuda())               # the matrix is left untouched
    ev2, ec2 = B.eigh_topk(Gt.clone(), k)
    assert torch.equal(ev, ev2) and torch.equal(ec, ec2)
    ev, X = ev.cpu().numpy(), ec.cpu().numpy().T
    w, v = np.linalg.eigh(G)
    w, v = w[::-1], v[:, ::-1]
    assert np.abs(ev - w[:k]).max() < 1e-12 * w[0] and np.all(np.diff(ev) <= 0)
    assert np.abs(X.T @ X - np.eye(k)).max() < 1e-12
    assert np.linalg.norm(G @ X - X * ev, axis=0).max() < 1e-12 * w[0]
    assert np.all(X[np.abs(X).argmax(axis=0), np.arange(k)] > 0)
    gap = (w[k - 1] - w[k]) / w[0]
    sin = np.linalg.svd(v[:, k:].T @ X, compute_uv=False).max()
    assert sin < max(1e-9, 1e-12 / gap), (sin, gap)


@pytest.mark.parametrize("n,k", [(800, 1), (800, 3), (1300, 4), (3000, 4)])
def test_eigh_fast_path_few_vectors(B, n, k):
    """k <= 4: the block used to be ONE 16-column tile, for which the product kernel has no instance (the 64-column kernel ran
    on it and faulted: svd_wrapper(..., ncomp=4) on more than 6144 frames, round 5); the block is now at least two tiles wide."""
    import torch
    G = _spectrum_matrix(n, (1.0 + np.arange(n)) ** -1.5, seed=n + k)
    got = B.eigh_topk_fast(torch.from_numpy(G).cuda(), k)
    assert got is not None
    ev, X = got[0].cpu().numpy(), got[1].cpu().numpy().T
    w = np.linalg.eigvalsh(G)[::-1]
    assert np.abs(ev - w[:k]).max() < 1e-12 * w[0]
    assert np.abs(X.T @ X - np.eye(k)).max() < 1e-12
    assert np.linalg.norm(G @ X - X * ev, axis=0).max() < 1e-12 * w[0]


def test_eigh_fast_path_gives_up_and_the_exact_path_takes_over(B):
    """A spectrum without a gap behind the k-th pair (flat bulk), a rank-deficient matrix and an indefinite one: the fast
    path reports why it stopped, leaves the matrix alone, and the call returns the exact solver's result."""
    import torch
    n, k = 700, 20
    ctx = B.get_context()
    ctx.set_option("eigh_fast", 1)
    cases = {"flat": (1.0 + 1e-4 * np.linspace(1, 0, n), 1), "rank 30": (np.r_[2.0 ** -np.arange(30.0), np.zeros(n - 30)], 3)}
    for name, (lam, why) in cases.items():
        G = _spectrum_matrix(n, lam, seed=5)
        ev, ec = B.eigh_topk(torch.from_numpy(G).cuda(), k)
        reason = ctx.get_option("eigh_fast_last_reason")
        assert reason == why, (name, reason)
        ctx.set_option("eigh_fast", 0)
        ev0, ec0 = B.eigh_topk(torch.from_numpy(G).cuda(), k)
        ctx.set_option("eigh_fast", 1)
        assert torch.equal(ev, ev0) and torch.equal(ec, ec0), name        # (bit-identical: the exact kernels ran on the same G)


def test_pca_with_and_without_the_fast_eigensolver(B):
    """pca() at a size where the fast path runs (n = 640 frames): frames with eigh_fast = 1 / 0 agree far inside the
    parity tolerance, and the asynchronous (pipelined) mode never takes the fast path (it reads Ritz values back)."""
    import torch
    from vip_amd.psfsub import pca
    from vip_amd.synth import synth_adi_device
    cube, ang = synth_adi_device(640, 64, seed=3)
    ang = np.linspace(0, 120, 640)
    ctx = B.get_context()
    ctx.set_option("eigh_fast", 1)
    f1 = pca(cube, ang, ncomp=12, verbose=False)
    assert ctx.get_option("eigh_fast_last_reason") == 0 and ctx.get_option("eigh_fast_last_locked") == 12
    ctx.set_option("eigh_fast", 0)
    f0 = pca(cube, ang, ncomp=12, verbose=False)
    ctx.set_option("eigh_fast", 1)
    ok = torch.isfinite(f0)
    assert torch.equal(ok, torch.isfinite(f1)) and float((f1[ok] - f0[ok]).abs().max()) < 2e-6


# ---- Gram matrix on the int8 matrix cores (gram_i8.hip) ------------------------------------------------------------

@pytest.mark.parametrize("n,P,batch", [(64, 4096, 1), (100, 10201, 1), (400, 70001, 1), (257, 131072, 1), (39, 8192, 5)])
def test_gram_on_the_int8_matrix_cores(B, n, P, batch):
    """Integer-slice (Ozaki) Gram: 5 digits -> 1e-10 of sqrt(G_ii G_jj) entry by entry (3e-12 of max|G| on image data), 6 digits
    -> float64 round-off; rows of very different magnitude, a zero row, negative data, ragged sizes; deterministic."""
    import torch
    rng = np.random.default_rng(n + P)
    M = rng.standard_normal((batch, n, P)).astype(np.float32) * np.logspace(-2, 2, P, dtype=np.float32)
    M[:, 1] = 0.0
    M[:, 2] *= np.float32(1e18)
    M[:, 3] *= np.float32(1e-18)
    M[:, 4, ::7] = 0.0
    ref = np.einsum("bip,bjp->bij", M.astype(np.float64), M.astype(np.float64))
    dg = np.sqrt(np.abs(np.einsum("bii->bi", ref)))
    scale = dg[:, :, None] * dg[:, None, :] + 1e-300
    Mt = torch.from_numpy(M).cuda()
    ctx = B.get_context()
    try:
        for mode, tol in ((1, 1e-10), (2, 2e-13)):
            ctx.set_option("gram_i8", mode)
            ctx.set_option("gram_i8_min_n", 16)
            G = (B.gram_batched(Mt) if batch > 1 else B.gram(Mt[0])[None])
            G2 = (B.gram_batched(Mt) if batch > 1 else B.gram(Mt[0])[None])
            assert torch.equal(G, G2)
            G = G.cpu().numpy()
            assert np.array_equal(G, np.swapaxes(G, 1, 2))
            assert np.all(G[:, 1] == 0.0)
            assert np.abs(G - ref).max() <= 1e-300 or (np.abs(G - ref) / scale).max() < tol, (mode, (np.abs(G - ref) / scale).max())
    finally:
        ctx.set_option("gram_i8", -1)
        ctx.set_option("gram_i8_min_n", 32)


def test_gram_int8_staging_variants_are_bit_identical(B):
    """Operands global -> LDS by DMA (default) or through registers, one or two LDS buffers, any slice count: the integer
    accumulation is exact, so every variant gives the same bits."""
    import torch
    rng = np.random.default_rng(11)
    M = torch.from_numpy(rng.standard_normal((200, 50000)).astype(np.float32) * 3.0).cuda()
    ctx = B.get_context()
    try:
        ctx.set_option("gram_i8", 1)
        ref = B.gram(M).clone()
        for dma, nbuf, slices in ((0, 2, 0), (0, 1, 0), (1, 2, 7), (0, 2, 7), (1, 2, 40)):
            ctx.set_option("gram_i8_dma", dma)
            ctx.set_option("gram_i8_nbuf", nbuf)
            ctx.set_option("gram_i8_slices", slices)
            G = B.gram(M)
            if slices == 0:
                assert torch.equal(G, ref), (dma, nbuf, slices)
            else:                       # other slices: other per-slice exponents and another summation order -- round-off only
                assert float((G - ref).abs().max()) < 1e-11 * float(ref.abs().max())
    finally:
        for o, v in (("gram_i8", -1), ("gram_i8_dma", 1), ("gram_i8_nbuf", 0), ("gram_i8_slices", 0)):
            ctx.set_option(o, v)


def test_gram_int8_non_finite_rows_and_the_default_rule(B):
    """A NaN / Inf sample poisons its row and column of G (as with the float64 kernel) and nothing else; by default the int8
    path serves large single problems only (>= 256 rows x 131072 samples) -- its result is within 1e-11 of the float64 kernel's."""
    import torch
    rng = np.random.default_rng(3)
    M = rng.standard_normal((300, 131072)).astype(np.float32)
    Mt = torch.from_numpy(M).cuda()
    ctx = B.get_context()
    try:
        ctx.set_option("gram_i8", 0)
        G0 = B.gram(Mt).cpu().numpy()
        ctx.set_option("gram_i8", -1)
        G1 = B.gram(Mt).cpu().numpy()
        assert not np.array_equal(G0, G1)                                   # the int8 path ran ...
        assert np.abs(G1 - G0).max() < 1e-11 * np.abs(G0).max()             # ... and agrees
        M[5, 77] = np.nan
        M[9, 100000] = np.inf
        ctx.set_option("gram_i8", 1)
        G = B.gram(torch.from_numpy(M).cuda()).cpu().numpy()
        bad = np.zeros(300, bool)
        bad[[5, 9]] = True
        assert np.all(~np.isfinite(G[bad])) and np.all(~np.isfinite(G[:, bad]))
        assert np.all(np.isfinite(G[~bad][:, ~bad]))
    finally:
        ctx.set_option("gram_i8", -1)


@pytest.mark.parametrize("nseg,n,m,k", [(2, 90, 64, 5), (3, 150, 120, 8), (2, 260, 200, 10), (1, 300, 230, 6)])
def test_annular_eigh_gathers_the_libraries_itself(B, nseg, n, m, k):
    """vipmi_annular_eigh_f64: the leading pairs of every library's sub-Gram matrix with the solver reading
    G[seg][idx[a]][idx[b]] itself -- bit-identical to vipmi_eigh_topk_f64 on the materialised, zero-padded matrices (the same
    values reach the same kernel); m = 230 is beyond the register-resident solver and takes the materialising route inside."""
    import torch
    rng = np.random.default_rng(nseg * 1000 + n)
    Gs, idx, ln, Hs = [], [], [], []
    for sg in range(nseg):
        M = rng.standard_normal((n, n + 40)) * (2.0 ** (-np.arange(n + 40) / 50.0))
        G = M @ M.T
        Gs.append(G)
        for j in range(n):
            lj = int(rng.integers(max(k, 3), m + 1)) if j % 7 else m
            ij = np.sort(rng.choice(n, size=lj, replace=False)).astype(np.int32)
            row = np.zeros(m, dtype=np.int32)
            row[:lj] = ij
            idx.append(row)
            ln.append(lj)
            H = np.zeros((m, m))
            H[:lj, :lj] = G[np.ix_(ij, ij)]
            Hs.append(H)
    Gs, idx, ln, Hs = np.stack(Gs), np.stack(idx), np.array(ln, dtype=np.int32), np.stack(Hs)
    dev = "cuda"
    Gt, it, lt = torch.from_numpy(Gs).to(dev), torch.from_numpy(idx).to(dev), torch.from_numpy(ln).to(dev)
    total = nseg * n
    work = torch.empty((total, m, m), dtype=torch.float64, device=dev)
    ev = torch.zeros((total, m), dtype=torch.float64, device=dev)
    ec = torch.zeros((total, m, m), dtype=torch.float64, device=dev)
    ctx = B.get_context()
    ctx.call("vipmi_annular_eigh_f64", B.ptr(Gt), nseg, n, B.ptr(it), B.ptr(lt), m, k, B.ptr(work), B.ptr(ev), B.ptr(ec))
    ev2, ec2 = B.eigh_topk(torch.from_numpy(Hs.copy()).to(dev), k, nact=lt)
    torch.cuda.synchronize()
    assert torch.equal(ev[:, :k], ev2) and torch.equal(ec[:, :k, :], ec2)
    for p in (0, total // 2, total - 1):
        lj = int(ln[p])
        w = np.linalg.eigvalsh(Hs[p][:lj, :lj])[::-1]
        np.testing.assert_allclose(ev[p, :k].cpu().numpy(), w[:k], atol=1e-12 * w[0])


@pytest.mark.parametrize("n,sizes,klen", [(70, (300, 1000, 37), 256), (200, (5000, 2049, 4096), 2048), (129, (513, 511), 512)])
def test_annular_fronts_of_all_segments_through_the_c_entries(B, n, sizes, klen):
    """vipmi_annular_gram_all_f32 / vipmi_annular_apply_all_f32 (round 6) called directly: ragged segments side by side (each padded
    to whole K-slices), their Gram matrices from ONE int8 product against float64 numpy, and the residual product written through the
    pixel list -- -1 entries (padding, pixels a later segment owns) untouched, the rest equal to (I - C) A of the per-segment entry."""
    import torch
    rng = np.random.default_rng(n + sum(sizes))
    side = int(np.ceil(np.sqrt(sum(sizes) * 1.3)))
    P = side * side
    cube = (rng.standard_normal((n, P)) * rng.uniform(0.2, 3.0, (1, P))).astype(np.float32)
    cube[:, ::17] *= 1e-3                                       # faint pixels beside bright ones
    perm = rng.permutation(P)
    segs, o = [], 0
    for sz in sizes:
        segs.append(np.sort(perm[o:o + sz]).astype(np.int32))
        o += sz - 5                                             # the next segment shares 5 pixels: the later one wins
    nseg = len(segs)
    offs = [0]
    for sg in segs:
        offs.append(offs[-1] + -(-sg.size // klen) * klen)
    Ptot = offs[-1]
    pix_all = np.full(Ptot, -1, dtype=np.int32)
    tile_seg = np.full(Ptot // 128, -1, dtype=np.int32)
    for si, sg in enumerate(segs):
        pix_all[offs[si]:offs[si] + sg.size] = sg
        tile_seg[offs[si] // 128:(offs[si] + sg.size + 127) // 128] = si
    pix_out = pix_all.copy()
    seen = set()
    for si in range(nseg - 1, -1, -1):                          # later segments first: they keep their pixels
        for j in range(offs[si], offs[si] + segs[si].size):
            if int(pix_all[j]) in seen:
                pix_out[j] = -1
            seen.add(int(pix_all[j]))
    seg_slice = (np.asarray(offs) // klen).astype(np.int32)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ct = up(cube)
    A_all = torch.empty((n, Ptot), dtype=torch.float32, device="cuda")
    G_all = torch.empty((nseg, n, n), dtype=torch.float64, device="cuda")
    ctx = B.get_context()
    pa_t, po_t, ts_t, ss_t = up(pix_all), up(pix_out), up(tile_seg), up(seg_slice)      # (named: they must outlive the enqueued launches)
    ctx.call("vipmi_annular_gram_all_f32", B.ptr(ct), n, P, B.ptr(pa_t), Ptot, klen, B.ptr(ss_t), nseg, B.ptr(A_all), B.ptr(G_all))
    Ah = A_all.cpu().numpy()
    for si, sg in enumerate(segs):
        Aseg = cube[:, sg]
        assert np.array_equal(Ah[:, offs[si]:offs[si] + sg.size], Aseg) and not Ah[:, offs[si] + sg.size:offs[si + 1]].any()
        Gref = Aseg.astype(np.float64) @ Aseg.astype(np.float64).T
        assert np.abs(G_all[si].cpu().numpy() - Gref).max() < 2e-11 * np.abs(Gref).max(), si
    # libraries: every frame's library = the frames at least 3 away, at most m of them; k components
    m, k = min(n - 8, 60), 4
    idx = np.zeros((nseg * n, m), dtype=np.int32)
    ln = np.zeros(nseg * n, dtype=np.int32)
    for p_ in range(nseg * n):
        j = p_ % n
        lib = np.array([f for f in range(n) if abs(f - j) >= 3][:m], dtype=np.int32)
        idx[p_, :lib.size] = lib
        ln[p_] = lib.size
    it, lt = up(idx), up(ln)
    work = torch.empty((nseg * n, m, m), dtype=torch.float64, device="cuda")
    ev = torch.zeros((nseg * n, m), dtype=torch.float64, device="cuda")
    ec = torch.zeros((nseg * n, m, m), dtype=torch.float64, device="cuda")
    ctx.call("vipmi_annular_eigh_f64", B.ptr(G_all), nseg, n, B.ptr(it), B.ptr(lt), m, k, B.ptr(work), B.ptr(ev), B.ptr(ec))
    out = torch.full((n, P), 7.0, dtype=torch.float32, device="cuda")
    kseg = up(np.full(nseg, k, dtype=np.int32))
    ctx.call("vipmi_annular_apply_all_f32", B.ptr(A_all), n, Ptot, B.ptr(ts_t), B.ptr(po_t), nseg, B.ptr(it), B.ptr(lt), m,
             B.ptr(G_all), B.ptr(ev), B.ptr(ec), B.ptr(kseg), k, P, B.ptr(out), None)
    oh = out.cpu().numpy()
    written = np.zeros(P, dtype=bool)
    kk = np.array([k], dtype=np.int32)
    for si, sg in enumerate(segs):
        # the per-segment entry on the same matrix, Gram matrix, eigenpairs: the same product, element by element
        Aseg = A_all[:, offs[si]:offs[si] + ((sg.size + 3) // 4) * 4].contiguous()
        R = torch.empty((1, n, Aseg.shape[1]), dtype=torch.float32, device="cuda")
        sl = slice(si * n, (si + 1) * n)
        its, lts, evs, ecs = it[sl].contiguous(), lt[sl].contiguous(), ev[sl].contiguous(), ec[sl].contiguous()
        ctx.call("vipmi_annular_apply_f32", B.ptr(Aseg), n, Aseg.shape[1], B.ptr(its), B.ptr(lts), m, m, B.ptr(G_all[si]), B.ptr(evs),
                 B.ptr(ecs), kk.ctypes.data_as(__import__("ctypes").c_void_p), 1, B.ptr(R))
        Rh = R[0].cpu().numpy()[:, :sg.size]
        own = pix_out[offs[si]:offs[si] + sg.size] >= 0
        assert np.array_equal(oh[:, sg[own]], Rh[:, own]), si
        written[sg[own]] = True
    assert written.sum() == len(seen) and np.all(oh[:, ~written] == 7.0)


@pytest.mark.parametrize("n,k", [(200, 150), (400, 330), (640, 300)])
def test_many_vectors_of_one_matrix_take_the_jacobi_kernel_with_the_exact_path_behind_it(B, n, k):
    """k > 0.4 n eigenpairs of one matrix (pca(ncomp = 200), a float ncomp whose CEVR asks for most of the spectrum): the one-sided
    Jacobi kernel first (every pair in ~10 ms at n = 400 where the tridiagonal solvers take 64 vectors at a time: 53 ms for 400),
    on a copy; when it does not converge within its sweep limit (forced here with eigh_max_sweeps = 1) the exact tridiagonal path
    runs on the untouched matrix.  Both against numpy."""
    import torch
    G = _spectrum_matrix(n, (1.0 + np.arange(n)) ** -1.2, seed=n + k)
    w, E = np.linalg.eigh(G)
    w, E = w[::-1], E[:, ::-1]
    ctx = B.get_context()
    for sweeps in (40, 1):
        ctx.set_option("eigh_max_sweeps", sweeps)
        try:
            ev, ec = B.eigh_topk(torch.from_numpy(G.copy()).cuda(), k)
        finally:
            ctx.set_option("eigh_max_sweeps", 40)
        ev, X = ev.cpu().numpy(), ec.cpu().numpy().T
        assert np.abs(ev - w[:k]).max() < 1e-12 * w[0], sweeps
        assert np.abs(X.T @ X - np.eye(k)).max() < 1e-10
        assert np.linalg.norm(G @ X - X * ev, axis=0).max() < 1e-11 * w[0], sweeps
    B.check_deferred()                      # the forced non-convergence left nothing latched behind


@pytest.mark.parametrize("n,k", [(200, 10), (120, 16), (96, 32), (150, 8)])
def test_eigh_batched_all_leading_values_from_one_wave(B, n, k):
    """Second launch of a big batch (>= 4 problems per CU, k >= 8): the k leading eigenvalues of every tridiagonal matrix come
    from ONE wave at once (tri::multisect_many, option eigh_many) -- same values as one wave per eigenvalue to the bracket width
    (2 eps), ragged and degenerate problems included; eigenpairs against numpy."""
    import torch
    rng = np.random.default_rng(n * 100 + k)
    batch = 1100
    X = rng.standard_normal((batch, n, 2 * n)) * (2.0 ** (-np.arange(2 * n) / 6.0))
    G = X @ X.transpose(0, 2, 1)
    G[5] = np.eye(n) * 3.0                                   # one eigenvalue n times
    G[6] = 0.0
    G[7] = np.diag(np.r_[np.ones(n // 2) * 2.0, np.ones(n - n // 2)])
    nact = rng.integers(1, n + 1, size=batch).astype(np.int32)
    nact[:8] = (1, 2, 3, k, n, n, n, n)
    for p in range(batch):
        G[p, nact[p]:, :] = 0
        G[p, :, nact[p]:] = 0
    ctx = B.get_context()
    res = {}
    try:
        for many in (0, 1):
            ctx.set_option("eigh_many", many)
            ev, E = B.eigh_topk(torch.from_numpy(G).cuda(), k, nact=torch.from_numpy(nact).cuda())
            torch.cuda.synchronize()
            res[many] = (ev.cpu().numpy(), E.cpu().numpy())
    finally:
        ctx.set_option("eigh_many", 1)
    scale = np.abs(res[0][0]).max(axis=1, keepdims=True) + 1e-300
    assert np.max(np.abs(res[0][0][:, :k] - res[1][0][:, :k]) / scale) < 1e-14
    ev, E = res[1]
    for p in list(range(8)) + list(range(8, batch, 97)):
        kk = min(k, int(nact[p]))
        w = np.linalg.eigvalsh(G[p])[::-1]
        sc = max(w[0], 1e-300)
        assert np.abs(ev[p, :kk] - w[:kk]).max() <= 1e-13 * sc
        V = E[p, :kk]
        assert np.abs(G[p] @ V.T - V.T * ev[p, :kk]).max() <= 1e-12 * sc
        assert np.abs(V @ V.T - np.eye(kk)).max() < 1e-12


def test_median_tile_order_options_are_bit_identical(B):
    """The median's tile -> XCD map (option median_xcd_chunk: chunks dealt round-robin, one range per XCD, plain order) only moves
    work between compute units: every variant returns the bits of np.nanmedian, on a cube whose edge pixels are the slow ones
    (part NaN, part a spike of near-zero values beside the real samples)."""
    import torch
    rng = np.random.default_rng(3)
    n, N = 400, 160
    x = rng.standard_normal((n, N, N)).astype(np.float32)
    yy, xx = np.mgrid[:N, :N]
    edge = (yy - N / 2) ** 2 + (xx - N / 2) ** 2 > (0.42 * N) ** 2
    x[:150, edge] *= 1e-3
    x[150:200, (yy + xx) % 7 == 0] = np.nan
    ref = np.nanmedian(x, axis=0)
    ctx = B.get_context()
    xt = torch.from_numpy(x).cuda()
    try:
        for ch in (-1, 0, 1, 2, 16, 64):
            ctx.set_option("median_xcd_chunk", ch)
            got = B.collapse(xt, "median").cpu().numpy()
            assert np.array_equal(got, ref, equal_nan=True), ch
    finally:
        ctx.set_option("median_xcd_chunk", -1)


def test_gather_scatter_round_trip_and_padding(B):
    """vipmi_gather_f32 / vipmi_scatter_f32 (annular fronts): A[f][j] = cube[f][pix[j]] with 0 in padding columns (pix < 0), and the
    scatter writes exactly the listed pixels -- frame counts that are not multiples of the kernels' frame groups included."""
    import torch
    rng = np.random.default_rng(11)
    for n, P, npx in ((13, 1000, 257), (400, 4096, 1024), (7, 50, 3)):
        cube = rng.standard_normal((n, P)).astype(np.float32)
        pix = rng.choice(P, size=npx, replace=False).astype(np.int32)
        pix[rng.random(npx) < 0.1] = -1
        ct, pt = torch.from_numpy(cube).cuda(), torch.from_numpy(pix).cuda()
        A = torch.full((n, npx), 7.0, device="cuda")
        ctx = B.get_context()
        ctx.call("vipmi_gather_f32", B.ptr(ct), n, P, B.ptr(pt), npx, B.ptr(A))
        want = np.where(pix[None, :] >= 0, cube[:, np.maximum(pix, 0)], 0.0).astype(np.float32)
        assert np.array_equal(A.cpu().numpy(), want)
        out = torch.full((n, P), -3.0, device="cuda")
        ctx.call("vipmi_scatter_f32", B.ptr(A), n, P, B.ptr(pt), npx, B.ptr(out))
        exp = np.full((n, P), -3.0, dtype=np.float32)
        exp[:, pix[pix >= 0]] = cube[:, pix[pix >= 0]]
        assert np.array_equal(out.cpu().numpy(), exp)


def test_cube_derotate_numpy_pipelined_is_bit_identical():
    """cube_derotate of a big numpy cube uploads, rotates and downloads blocks of frames at the same time (uploader / downloader
    threads on two copy streams): same kernels on the same frames -- identical to the one-shot path for float32 and float64 input
    (dtype kept), NaN pixels and angles in every quadrant included."""
    import os
    from vip_amd.preproc import cube_derotate
    rng = np.random.default_rng(8)
    for n, N, dt in ((130, 384, np.float32), (70, 512, np.float64), (64, 511, np.float32)):
        cube = rng.standard_normal((n, N, N)).astype(dt)
        cube[:, :4, :7] = np.nan
        ang = np.linspace(-179, 178, n)
        res = {}
        try:
            for h in ("0", "1"):
                os.environ["VIPMI_HOSTIN"] = h
                res[h] = cube_derotate(cube, ang)
        finally:
            os.environ.pop("VIPMI_HOSTIN", None)
        assert res["1"].dtype == dt and res["0"].dtype == dt
        assert np.array_equal(res["0"], res["1"], equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("with_std", [0, 1])
def test_spatial_scaling_of_a_float64_matrix_through_the_c_entries(B, with_std):
    """vipmi_spat_center_f64 / vipmi_gram_offset_u_f64 (round 6) called directly on a matrix of detector counts with zero padding:
    D + u mu^T reproduces sklearn's scale(M, axis=1) (var/shapes.py:740-781) formed in float64 numpy to float32 rounding of the SMALL
    matrix D, the padding stays zero, and D D^T plus the offset terms is the Gram matrix of the scaled matrix."""
    import torch
    rng = np.random.default_rng(40 + with_std)
    n, Preal, P = 37, 1003, 1008
    M = np.zeros((n, P))
    M[:, :Preal] = 7000.0 + 45.0 * rng.standard_normal((n, Preal)) * rng.uniform(0.5, 2.0, (n, 1)) + rng.uniform(-300, 300, (1, Preal))
    ctx = B.get_context()
    Mt = torch.from_numpy(M).cuda()
    D = torch.full((n, P), 9.0, dtype=torch.float32, device="cuda")
    mu = torch.empty((P,), dtype=torch.float64, device="cuda")
    mu32 = torch.empty((P,), dtype=torch.float32, device="cuda")
    u = torch.empty((n,), dtype=torch.float64, device="cuda")
    ctx.call("vipmi_spat_center_f64", B.ptr(Mt), n, P, Preal, with_std, B.ptr(D), B.ptr(mu), B.ptr(mu32), B.ptr(u))
    torch.cuda.synchronize()
    X = M[:, :Preal]
    m = X.mean(axis=1, keepdims=True)
    sd = X.std(axis=1, keepdims=True) if with_std else np.ones_like(m)
    want = (X - m) / sd
    Dh, muh, uh = D.cpu().numpy().astype(np.float64), mu.cpu().numpy(), u.cpu().numpy()
    assert np.allclose(uh, 1.0 / sd[:, 0], rtol=1e-13)
    assert not Dh[:, Preal:].any() and not muh[Preal:].any()
    got = Dh[:, :Preal] + uh[:, None] * muh[None, :Preal]
    small = np.abs(want - uh[:, None] * muh[None, :Preal]).max()           # the size of D: what float32 has to hold
    assert np.abs(got - want).max() <= 2.0 ** -23 * small
    assert np.array_equal(mu32.cpu().numpy(), muh.astype(np.float32))
    G = torch.empty((n, n), dtype=torch.float64, device="cuda")
    ctx.call("vipmi_gram_f32", B.ptr(D), n, P, P, B.ptr(G))
    ctx.call("vipmi_gram_offset_u_f64", B.ptr(D), B.ptr(mu), B.ptr(u), n, P, B.ptr(G))
    torch.cuda.synchronize()
    Gw = got @ got.T
    assert np.abs(G.cpu().numpy() - Gw).max() <= 1e-11 * np.abs(Gw).max()


@pytest.mark.gpu
@pytest.mark.parametrize("n,batch", [(82, 1), (250, 6), (259, 3), (513, 2)])
def test_jacobi_converges_on_clusters_and_rank_deficient_matrices(B, n, batch):
    """vipmi_eigh_f64 (one-sided Jacobi) on spectra it did not converge on before round 6 (tools/hunt_eigh_sizes.py: 'did not
    converge in 40 sweeps', nor in 300): a cluster of eigenvalues that agree to nine digits -- the rotation angle was steered by
    beta - alpha formed in float32 -- and rank-deficient Gram matrices, whose null columns are rounding noise that no rotation
    orthogonalises to a relative tolerance.  Eigenvalues to 1e-12 of the largest, the vectors of the live eigenvalues orthonormal with
    small residuals."""
    import torch
    rng = np.random.default_rng(n)
    for shape in ("cluster", "rank"):
        Gs = []
        for _ in range(batch):
            if shape == "cluster":
                Q, _r = np.linalg.qr(rng.standard_normal((n, n)))
                lam = np.concatenate([np.full(n // 2, 5.0) + 1e-9 * rng.standard_normal(n // 2), rng.uniform(0.1, 1.0, n - n // 2)])
                X = Q * np.sqrt(lam)
            else:
                X = rng.standard_normal((n, n // 3))
            Gs.append(X @ X.T)
        G = np.stack(Gs)
        ev, ec = B.eigh(torch.from_numpy(G).cuda())
        B.check_deferred()
        print("jacobi %s n %d batch %d: %d sweeps (problem 0)" % (shape, n, batch, B.get_context().get_option("eigh_last_sweeps")))
        ev, ec = ev.cpu().numpy(), ec.cpu().numpy()
        for b in range(batch):
            w = np.linalg.eigvalsh(G[b])[::-1]
            assert np.abs(ev[b] - w).max() < 1e-12 * w[0], (shape, b)
            live = ev[b] > 1e-11 * w[0]
            V = ec[b][live]
            assert np.abs(V @ V.T - np.eye(int(live.sum()))).max() < 1e-10, (shape, b)
            assert np.abs(G[b] @ V.T - V.T * ev[b][live]).max() < 1e-10 * w[0], (shape, b)
            assert np.abs((ec[b] ** 2).sum(1) - 1).max() < 1e-12


@pytest.mark.gpu
def test_batched_many_vector_request_falls_back_per_problem_when_jacobi_gives_up(B):
    """More than 64 vectors per matrix in a batch of more than four (a 4-D cube with ncomp = 100) has only the one-sided Jacobi kernel
    to serve every problem at once; spectra graded over twelve decades do not converge within its sweep limit (NOTES round 6).  In
    synchronous mode the problems it gives up on are solved one by one by the matrix-in-L2 tridiagonal solver from a copy: the call
    returns the eigenpairs instead of 'did not converge', nothing stays latched for check_deferred."""
    import torch
    rng = np.random.default_rng(12)
    n, k, batch = 300, 90, 6
    Gs = []
    for b in range(batch):
        Q, _r = np.linalg.qr(rng.standard_normal((n, n)))
        decades = 12 if b % 2 == 0 else 4                      # every other problem is one Jacobi handles
        lam = 10.0 ** (-decades * np.arange(n) / n)
        G = (Q * lam) @ Q.T
        Gs.append(0.5 * (G + G.T))
    G = np.stack(Gs)
    ev, ec = B.eigh_topk(torch.from_numpy(G).cuda(), k)
    B.check_deferred()
    assert B.get_context().get_option("eigh_batch_fallback") == 3
    ev, ec = ev.cpu().numpy(), ec.cpu().numpy()
    for b in range(batch):
        w = np.linalg.eigvalsh(G[b])[::-1]
        assert np.abs(ev[b, :k] - w[:k]).max() < 1e-12 * w[0], b
        V = ec[b, :k]
        assert np.abs(V @ V.T - np.eye(k)).max() < (1e-9 if b % 2 == 0 else 1e-7), b
        assert np.abs(G[b] @ V.T - V.T * ev[b, :k]).max() < 1e-9 * w[0], b
