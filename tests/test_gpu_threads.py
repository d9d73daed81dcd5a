""""thread-safe across handles" -- several host threads, each with its own stream / context, issuing SYNCHRONOUS
calls on one GPU must get the single-thread results bit for bit.  Round 3 shipped a violation (wrong medians beside another
context's int8 Gram product); the cause was a gfx950 packed-FP32 operand form that goes wrong beside the 16x16x64 int8 MFMA of
another wave (vip_amd/csrc/common.h VIPMI_NO_PK32, tools/hunt/): these tests are the regression net for the whole class, every
stage beside every other stage on the library's DEFAULT options (int8 Gram on)."""
import threading

import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    import torch
    return torch.equal(torch.nan_to_num(a, nan=1234.5), torch.nan_to_num(b, nan=1234.5))


@pytest.fixture(scope="module")
def c2():
    import torch
    from vip_amd.synth import synth_adi_device
    ct, ang = synth_adi_device(400, 512, seed=0)
    torch.cuda.synchronize()
    return ct, ang


def test_three_threads_of_synchronous_pca_at_c2_on_defaults(c2):
    """BASELINE configs[1] from three threads at once, library defaults: every frame identical to the single-thread frame."""
    import torch
    from vip_amd.psfsub import pca
    ct, ang = c2
    ref = pca(ct, ang, ncomp=20, verbose=False, check_memory=False).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(ref).all()
    errs = []

    def work(k):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for i in range(12):
                    o = pca(ct, ang, ncomp=20, verbose=False, check_memory=False)
                    torch.cuda.current_stream().synchronize()
                    if not _same(o, ref):
                        errs.append((k, i, int((o != ref).sum()), float((o - ref).abs().nan_to_num(nan=9e9).max())))
        except Exception as e:          # noqa: BLE001
            errs.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs[:5]


STAGES = ("gram", "eigh", "project", "derotate", "median", "mean", "pca")


@pytest.fixture(scope="module")
def stage_fns(c2):
    import torch
    from vip_amd import backend as B
    from vip_amd.psfsub import pca
    ct, ang = c2
    M = ct.reshape(400, -1)
    G0 = B.gram(M).clone()
    torch.cuda.synchronize()

    def st_eigh():
        ev, ec = B.eigh_topk(G0.clone(), 20)
        return torch.cat([ev.flatten(), ec.flatten()])

    return {"gram": lambda: B.gram(M), "eigh": st_eigh, "project": lambda: B.pca_project(M, 20)[0],
            "derotate": lambda: B.derotate(ct, ang), "median": lambda: B.collapse(ct, "median"),
            "mean": lambda: B.collapse(ct, "mean"), "pca": lambda: pca(ct, ang, ncomp=20, verbose=False, check_memory=False)}


@pytest.mark.parametrize("load", ("gram", "project", "derotate", "median", "eigh"))
def test_every_stage_beside_two_threads_of_one_stage(stage_fns, load):
    """tools/stage_threads.py at reduced repetitions: stage X, checked call by call against its own single-thread result, while
    two other threads loop stage `load` on their own streams (round 3: median beside gram differed in 40 of 40 calls)."""
    import torch
    stop, lerr = [False], []

    def loader():
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                while not stop[0]:
                    stage_fns[load]()
                    torch.cuda.current_stream().synchronize()
        except Exception as e:          # noqa: BLE001
            lerr.append(repr(e))

    refs = {}
    for chk in STAGES:
        refs[chk] = stage_fns[chk]().clone()
    torch.cuda.synchronize()
    tl = [threading.Thread(target=loader) for _ in range(2)]
    [t.start() for t in tl]
    bad = {}
    try:
        with torch.cuda.stream(torch.cuda.Stream()):
            for chk in STAGES:
                for i in range(6):
                    o = stage_fns[chk]()
                    torch.cuda.current_stream().synchronize()
                    if not _same(o, refs[chk]):
                        bad.setdefault(chk, []).append(i)
    finally:
        stop[0] = True
        [t.join() for t in tl]
    assert not lerr, lerr[:2]
    assert not bad, "stages that differ beside two threads of %s: %s" % (load, bad)


def test_small_and_odd_sized_cubes_from_threads():
    """The generic (non power-of-two) rotation kernels and the small-cube paths beside int8 Gram products of a big cube."""
    import numpy as np
    import torch
    from vip_amd import backend as B
    from vip_amd.psfsub import pca
    from vip_amd.synth import synth_adi_device
    big, _ = synth_adi_device(400, 512, seed=3)
    Mb = big.reshape(400, -1)
    cases = []
    for n, N, k in ((40, 101, 5), (60, 128, 6), (30, 255, 3)):
        ct, ang = synth_adi_device(n, N, seed=n)
        cases.append((ct, ang, k, pca(ct, ang, ncomp=k, verbose=False, check_memory=False).clone()))
    torch.cuda.synchronize()
    stop = [False]

    def loader():
        with torch.cuda.stream(torch.cuda.Stream()):
            while not stop[0]:
                B.gram(Mb)
                torch.cuda.current_stream().synchronize()

    tl = [threading.Thread(target=loader) for _ in range(2)]
    [t.start() for t in tl]
    bad = []
    try:
        with torch.cuda.stream(torch.cuda.Stream()):
            for rep in range(10):
                for ct, ang, k, ref in cases:
                    o = pca(ct, ang, ncomp=k, verbose=False, check_memory=False)
                    torch.cuda.current_stream().synchronize()
                    if not _same(o, ref):
                        bad.append((tuple(ct.shape), rep, float(np.nanmax(np.abs((o - ref).cpu().numpy())))))
    finally:
        stop[0] = True
        [t.join() for t in tl]
    assert not bad, bad[:5]


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("0", "150"), ("1", "240", "4")])
def test_repeated_calls_in_random_order_reproduce_their_first_results(args):
    """State between calls (plan / table / library caches, workspaces; tools/hunt_call_sequences.py): twenty call configurations that
    share sizes but not data and angle lists of equal length but different values, each run once and then repeated in random order --
    interleaved with workspace releases, fresh copies of the inputs and cuda inputs; on one thread and on four threads with a stream
    each -- bit for bit the first result every time."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cp = subprocess.run([sys.executable, os.path.join(root, "tools", "hunt_call_sequences.py")] + list(args), capture_output=True, text=True,
                        timeout=600)
    assert cp.returncode == 0, cp.stderr[-2000:]
    assert "failures: 0" in cp.stdout, cp.stdout[-3000:]
