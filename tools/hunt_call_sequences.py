"""State between calls: caches of plans, uploaded tables, annulus libraries, workspaces.  Twenty call configurations (pca, pca_annular,
median_sub, cube_derotate; shapes that share sizes but not data, angle lists of equal length but different values, parameters that
share a plan key but not a result) are each run once, then 150 times more in random order -- interleaved with workspace releases,
fresh copies of the inputs (same values, other objects), float64 / cuda variants -- and every repeat must reproduce its first result
bit for bit.   python tools/hunt_call_sequences.py [seed [repeats [threads]]]
threads > 1: the repeats run on that many host threads at once, each under its own torch stream (a context per stream)."""
import sys, os, time, gc, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.psfsub import pca, pca_annular, median_sub
from vip_amd.preproc import cube_derotate
from vip_amd.synth import synth_adi

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 150
nthreads = int(sys.argv[3]) if len(sys.argv) > 3 else 1
import threading
rng = np.random.default_rng(31000 + seed)
shapes = [(24, 48), (24, 48), (40, 63), (40, 63), (61, 101), (100, 128), (100, 128), (37, 130)]
cubes = [synth_adi(n, N, seed=100 * seed + i)[0].astype(np.float32) for i, (n, N) in enumerate(shapes)]
def angles(n, j):
    return [np.linspace(0, 80, n), np.linspace(0, 81, n), np.linspace(-40, 120, n), np.sort(np.random.default_rng(j).uniform(0, 150, n))][j % 4]
configs = []
for ci in range(20):
    c = int(rng.integers(len(cubes))); n, N = shapes[c]
    ang = angles(n, int(rng.integers(4)))
    kind = int(rng.integers(6))
    if kind == 0:
        kw = dict(ncomp=int(rng.integers(1, 8)), scaling=(None, "temp-mean", "spat-standard")[rng.integers(3)])
        if rng.integers(3) == 0:
            kw["mask_center_px"] = int(rng.integers(2, 6))
        fn = ("pca", kw)
    elif kind == 1:
        fn = ("pca_annular", dict(ncomp=int(rng.integers(1, 5)), asize=int(rng.integers(5, 9)), fwhm=4, delta_rot=(0.2, float(rng.choice([0.6, 0.8, 1.0]))),
                                  n_segments=int(rng.integers(1, 3)), radius_int=int(rng.integers(0, 4))))
    elif kind == 2:
        fn = ("median_sub", dict(mode="annular", asize=int(rng.integers(4, 8)), fwhm=4, delta_rot=float(rng.choice([0.5, 1.0])), nframes=4)) if N >= 60 else ("median_sub", {})
    elif kind == 3:
        fn = ("cube_derotate", {})
    elif kind == 4:
        fn = ("pca", dict(ncomp=(1, 5, 2), full_output=True))           # grid of PCs -> (frames, list)
    else:
        fn = ("pca", dict(ncomp=int(rng.integers(1, 6)), cube_ref=cubes[(c + 1) % len(cubes)] if shapes[(c + 1) % len(cubes)][1] == N else None))
    configs.append((c, ang, fn))


def run(cfg, variant):
    c, ang, (name, kw) = cfg
    cube = cubes[c]
    if variant == 1:
        cube, ang = cube.copy(), ang.copy()                  # other objects, same values
    elif variant == 2 and name != "median_sub":
        cube = torch.from_numpy(cube).cuda()
    kw = dict(kw)
    if name == "pca":
        out = pca(cube, ang, verbose=False, **kw)
    elif name == "pca_annular":
        out = pca_annular(cube, ang, verbose=False, **kw)
    elif name == "median_sub":
        out = median_sub(cube, ang, verbose=False, **kw)
    else:
        out = cube_derotate(cube, ang)
    if isinstance(out, tuple):
        out = out[0]
    if hasattr(out, "cpu"):
        out = out.cpu().numpy()
    return np.asarray(out, dtype=np.float64)


first = []
for i, cfg in enumerate(configs):
    first.append(run(cfg, 0))
bad = 0
t0 = time.time()
lock = threading.Lock()


def worker(tid, nrep):
    global bad
    lrng = np.random.default_rng(41000 + 97 * seed + tid)
    stream = torch.cuda.Stream() if nthreads > 1 else torch.cuda.current_stream()
    with torch.cuda.stream(stream):
        for r in range(nrep):
            i = int(lrng.integers(len(configs)))
            variant = int(lrng.integers(3))
            ev = int(lrng.integers(12))
            if ev == 0 and nthreads == 1:
                B.release_workspaces()
            elif ev == 1:
                gc.collect()
            elif ev == 2 and nthreads == 1:
                torch.cuda.empty_cache()
            try:
                out = run(configs[i], variant)
                same = out.shape == first[i].shape and np.array_equal(np.nan_to_num(out, nan=123.0), np.nan_to_num(first[i], nan=123.0))
                if not same:
                    d = np.nanmax(np.abs(out - first[i])) if out.shape == first[i].shape else -1
                    with lock:
                        bad += 1
                    print("FAIL thread %d repeat %d config %d %s (variant %d, event %d): differs from its first result, max|d| %.3e" % (tid, r, i, configs[i][2], variant, ev, d), flush=True)
            except Exception as e:
                with lock:
                    bad += 1
                print("FAIL thread %d repeat %d config %d %s: %s" % (tid, r, i, configs[i][2], "".join(traceback.format_exception_only(type(e), e)).strip()[:300]), flush=True)
        B.check_deferred()
        stream.synchronize()


if nthreads == 1:
    worker(0, repeats)
else:
    ths = [threading.Thread(target=worker, args=(t, repeats // nthreads)) for t in range(nthreads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
B.check_deferred()
print("seed %d: %d repeats of %d configurations on %d thread(s) in %.0f s, failures: %d" % (seed, repeats, len(configs), nthreads, time.time() - t0, bad))
