#!/usr/bin/env python
"""Benchmark of the ADI PSF-subtraction hot path on MI355X (BASELINE.json metric:
"ADI cube frames/sec (and ms/SVD) at ncomp=20, 400x512x512").

    python bench.py --gpus N --steps K --warmup W

One "step" = one full `vip_amd.psfsub.pca(cube, angles, ncomp=20)` call (Gram -> eigensolver ->
project/subtract -> 3-shear FFT derotation -> median collapse -> D2H of the final frame) on a synthetic
400x512x512 float32 cube that is already resident in HBM.  N > 1 (launched with torch.distributed.run,
one rank per GPU): every rank processes its own cube -- the "survey mode" sharding of SURVEY.md 8(e):
no data-path collective, only the barrier / max-over-ranks timing -- so scaling is weak.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     : dominant kernel (rs_shear2, the column shear of the real-split FFT derotation) -- algorithmic bytes of
                 the derotation stage per launch / its average launch duration (hipEvent pairs recorded around
                 the kernel on the stream it is launched on, inside the timed region)
  cpu_baseline : the numpy oracle (oracle/ref_cpu.py, a port of the reference's svd_mode='lapack' +
                 imlib='vip-fft' + nanmedian path) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import gc
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_VALU_PEAK_TF = 157.3


def _derotate_one(args):
    from oracle import ref_cpu as O
    frame, angle = args
    return float(np.nansum(O.cube_derotate(frame[None], np.array([angle]))))


def cpu_baseline(n, N, k, frames_1t=16):
    """Oracle (numpy port of the reference's svd_mode='lapack' + imlib='vip-fft' + nanmedian path) on a bounded sample of
    the same workload, extrapolated linearly to the full cube: t = t_svd_project(full matrix, BLAS threads) +
    n * t_derotate(1 frame) + t_median.  Two figures: ``value`` = the derotation spread over all
    host cores by a process pool (what the reference's nproc=<cores> does), ``value_nproc1`` = nproc=1 semantics
    (derotation on one thread, 16 frames timed)."""
    import multiprocessing as mp
    from oracle import ref_cpu as O
    from vip_amd.synth import synth_adi
    cores = os.cpu_count() or 1
    cube, angles = synth_adi(n, N, seed=0)
    t0 = time.perf_counter()
    res = O.project_subtract(cube, k, None, None, "lapack")
    t_svd = time.perf_counter() - t0
    frames_1t = min(frames_1t, n)
    t0 = time.perf_counter()
    O.cube_derotate(res[:frames_1t], angles[:frames_1t])
    t_rot1 = (time.perf_counter() - t0) / frames_1t
    # all cores: one frame per task, 2 tasks per worker (fork: the workers inherit numpy / the oracle already imported)
    nw = min(cores, n)
    npool = min(n, 2 * nw)
    t_rot_pool = None
    try:
        with mp.get_context("fork").Pool(nw) as pool:
            pool.map(_derotate_one, [(res[i], angles[i]) for i in range(nw)])        # warm the workers
            t0 = time.perf_counter()
            pool.map(_derotate_one, [(res[i], angles[i]) for i in range(npool)], chunksize=1)
            t_rot_pool = (time.perf_counter() - t0) / npool
    except Exception as e:                      # (a box that forbids fork: report the one-thread figure only)
        sys.stderr.write("cpu_baseline: process pool failed (%s)\n" % e)
    step = 8
    t0 = time.perf_counter()
    O.cube_collapse(res[:, ::step, :], "median")
    t_med = (time.perf_counter() - t0) * step
    total1 = t_svd + n * t_rot1 + t_med
    total = t_svd + n * (t_rot_pool if t_rot_pool is not None else t_rot1) + t_med
    return {"value": n / total, "unit": "frames/s", "cores": int(cores if t_rot_pool is not None else 1), "kind": "port",
            "value_nproc1": n / total1,
            "sample": "full %dx%dx%d SVD+project (%.1fs, BLAS threads) + derotation of %d frames on 1 thread "
                      "(%.3fs/frame) and of %d frames over a pool of %d processes (%s s/frame effective) + median on "
                      "1/%d of the pixels (%.1fs scaled); extrapolated to the full cube: %.0fs all cores, %.0fs nproc=1"
                      % (n, N, N, t_svd, frames_1t, t_rot1, npool, nw,
                         "%.4f" % t_rot_pool if t_rot_pool is not None else "n/a", step, t_med, total, total1),
            "ms_per_svd": 1e3 * t_svd}


def pmc_traffic(frames_per_launch, N, kernel="rs_shear2"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/rNN_pmc_hbm.json, written by tools/pmc_summary.py: FETCH_SIZE and WRITE_SIZE collected in separate
    --pmc runs of one 400x512x512 pca() call).  The kernel reads 64-byte row segments (8 B/lane), for which the
    gfx950 FETCH_SIZE counter needs no x2 correction (MI355X_MICROARCH.md, HBM section); scaled per frame.
    None if absent / other frame size."""
    import glob
    if N != 512:
        return None
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json"))
                   if re.fullmatch(r"r\d+_pmc_hbm\.json", os.path.basename(f)))      # the round's final pass (rNN), not interim ones
    if not files:
        return None
    try:
        doc = json.load(open(files[-1]))
        for name, e in doc["kernels"].items():
            if name.startswith(kernel):
                per_launch = 1024.0 * (e["FETCH_SIZE_KB_per_launch"] + e["WRITE_SIZE_KB_per_launch"])
                if "frames_per_launch" in doc:          # (round 2 on) every profiled launch covers the whole cube
                    return per_launch * frames_per_launch / float(doc["frames_per_launch"])
                return per_launch * frames_per_launch / (400.0 / e["launches"])
    except Exception:
        return None
    return None


class PowerSampler:
    """Socket power and shader clock of THIS rank's GPU from sysfs hwmon (power1_input / power1_average in uW, freq1_input in Hz),
    sampled by a host thread every 20 ms while a leg runs.  The card is matched by PCI address; the chip-filling kernels of the
    path run AT the board's power cap (1381 of 1400 W under the shears, shader clock 1.96 of 2.4 GHz), which is what
    bounds them -- the roofline fractions below are to be read with that in mind."""

    def __init__(self, device):
        import glob
        self.dir = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(device)
            want = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            want = None
        cands = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
        for d in cands:
            try:
                real = os.path.realpath(os.path.join(d, "..", ".."))
            except Exception:
                continue
            if want and want in real:
                self.dir = d
        self.cands = cands if self.dir is None else [self.dir]
        self.samples, self._stop, self._th = [], False, None

    def _read(self, d):
        out = {}
        for name in ("power1_input", "power1_average", "freq1_input", "power1_cap"):
            try:
                out[name] = int(open(os.path.join(d, name)).read())
            except Exception:
                pass
        return out

    def _loop(self):
        while not self._stop:
            self.samples.append([self._read(d) for d in self.cands])
            time.sleep(0.02)

    def start(self):
        import threading
        if not self.cands:
            return self
        self.samples, self._stop = [], False
        self._th = threading.Thread(target=self._loop, daemon=True)
        self._th.start()
        return self

    def stop(self, what):
        if self._th is None:
            return None
        self._stop = True
        self._th.join()
        self._th = None
        smp = self.samples[2:-1] if len(self.samples) > 6 else self.samples        # (drop the ramp at both ends)
        if not smp:
            return None
        best = None
        for i in range(len(self.cands)):                 # unmatched card: the one that draws the most (ours is the busy one)
            pw = [s[i].get("power1_input", s[i].get("power1_average")) for s in smp]
            pw = [v for v in pw if v is not None]
            if not pw:
                continue
            fr = [s[i]["freq1_input"] for s in smp if "freq1_input" in s[i]]
            rec = {"mean_w": float(np.mean(pw)) / 1e6, "max_w": float(np.max(pw)) / 1e6,
                   "cap_w": (smp[-1][i].get("power1_cap") or 0) / 1e6 or None,
                   "sclk_mhz_mean": (float(np.mean(fr)) / 1e6 if fr else None), "samples": len(pw), "during": what,
                   "card_matched_by_pci": self.dir is not None}
            if best is None or rec["mean_w"] > best["mean_w"]:
                best = rec
        return best


def numpy_in_legs(n, N, k, angles, seed0, pca, B, torch):
    """The reference's callers pass numpy arrays (metrics/contrcurve.py:768-790, psfsub/pca_fullfr.py:137): the drop-in
    signature pays the PCIe upload that `value` excludes ("H2D ... reported separately").  Measured here:
    the raw host->device copy of one cube (pageable memory, as a caller's array is), one synchronous pca(numpy cube) ->
    numpy frame, and pca_many() over distinct numpy cubes (the upload of cube i+1 overlaps the kernels of cube i)."""
    from vip_amd.psfsub.pca_fullfr import pca_many
    from vip_amd.synth import synth_adi
    hosts = [synth_adi(n, N, seed=seed0 + 100 + d)[0] for d in range(3)]
    torch.cuda.synchronize()
    t = B.to_device_f32(hosts[0]); torch.cuda.synchronize(); del t
    t0 = time.perf_counter()
    for h in hosts:
        t = B.to_device_f32(h)
        torch.cuda.synchronize()
        del t
    h2d_ms = (time.perf_counter() - t0) / len(hosts) * 1e3
    pca(hosts[0], angles, ncomp=k, verbose=False, check_memory=False)
    t0 = time.perf_counter()
    reps = 6
    for i in range(reps):
        fr = pca(hosts[i % len(hosts)], angles, ncomp=k, verbose=False, check_memory=False)
    lat_ms = (time.perf_counter() - t0) / reps * 1e3
    assert isinstance(fr, np.ndarray) and np.isfinite(fr[N // 2 - 4:N // 2 + 4, N // 2 - 4:N // 2 + 4]).all()
    many = [hosts[i % len(hosts)] for i in range(12)]
    pca_many(many[:3], [angles] * 3, ncomp=k, check_memory=False)
    t0 = time.perf_counter()
    outs = pca_many(many, [angles] * len(many), ncomp=k, check_memory=False)
    many_ms = (time.perf_counter() - t0) / len(many) * 1e3
    assert len(outs) == len(many) and isinstance(outs[0], np.ndarray)
    # full_output=True (the reference's most common diagnostic call, pca_fullfr.py:759-793): frame, pcs and three cube-sized arrays
    # back to numpy -- into pinned blocks the arrays own (backend.to_host_many), reused once the caller drops the previous results
    fo = pca(hosts[0], angles, ncomp=k, verbose=False, check_memory=False, full_output=True)
    fo_first_bytes = sum(a.nbytes for a in fo)
    del fo
    t0 = time.perf_counter()
    for i in range(4):
        fo = pca(hosts[i % len(hosts)], angles, ncomp=k, verbose=False, check_memory=False, full_output=True)
        assert isinstance(fo[3], np.ndarray) and fo[3].shape == hosts[0].shape
        del fo
    fo_ms = (time.perf_counter() - t0) / 4 * 1e3
    # a float64 cube (twice the bytes over PCIe; the float64 route: temporal mean carried in float64, csrc/pca_f64.hip)
    c64 = hosts[0].astype(np.float64)
    pca(c64, angles, ncomp=k, verbose=False, check_memory=False)
    t0 = time.perf_counter()
    for _ in range(3):
        f64 = pca(c64, angles, ncomp=k, verbose=False, check_memory=False)
    f64_ms = (time.perf_counter() - t0) / 3 * 1e3
    assert f64.dtype == np.float64
    del c64
    gb = hosts[0].nbytes / 1e9
    return {"h2d_ms": h2d_ms, "h2d_gbs": gb / (h2d_ms * 1e-3), "host_memory": "pageable (a caller's numpy array)",
            "float64_latency_ms_per_call": f64_ms,
            "full_output_latency_ms_per_call": fo_ms, "full_output_bytes_to_host": fo_first_bytes,
            "full_output_note": "pca(numpy cube, full_output=True) -> five numpy arrays (1.26 GB at C2): the PCIe floor is upload + "
                                "download of four cubes; results land in pinned blocks that the arrays own (round 5: t.cpu() into "
                                "fresh pageable pages, 25-44 ms per array)",
            "latency_ms_per_call": lat_ms, "value": n / (lat_ms * 1e-3), "unit": "frames/s",
            "pipelined": {"value": n / (many_ms * 1e-3), "ms_per_cube": many_ms, "cubes": len(many),
                          "note": "pca_many: the (synchronous, pageable) upload of cube i+1 runs beside the kernels of cube i; "
                                  "bounded by PCIe: %.1f ms per %.0f MB cube" % (h2d_ms, hosts[0].nbytes / 1e6)},
            "note": "one synchronous call = upload + the serial call: everything after the Gram needs the Gram, and the Gram needs "
                    "every frame -- but its tile (i, j) only the row blocks i and j, so the library uploads the cube in blocks of 64 "
                    "frames and forms the Gram of the blocks that have arrived under the copy (vipmi_pca_fullframe_hostin_f32: "
                    "~0.4 ms of the 0.7 ms Gram stage hidden, bit-identical; DESIGN 7.8)"}


def shape_legs(pca, B, torch, c2_latency_ms):
    """The reference's OWN shapes (round-5 VERDICT #2): its frames are odd-sized by convention (`frame_center`, metrics/contrcurve.py:725
    insists on odd PSFs, fm/fakecomp.py:704-711 forces odd sizes), and an odd N makes the padded period Le = 4 N a non-power of two:
    the derotation then runs as power-of-two circular convolutions (csrc/derotate_conv.inc) instead of the plain transforms of the
    BASELINE shapes.  (a) 61 x 101 x 101, ncomp = 5 -- the shape of the only timing the reference publishes for `pca`
    (docs/source/tutorials/01A_quickstart.ipynb cell 53: 2.37 s on an unspecified laptop => 25.7 frames/s);
    (b) 400 x 511 x 511, ncomp = 20 beside C2.  One synchronous call each, cube resident (the metric's definition) and numpy in."""
    from vip_amd.psfsub.pca_fullfr import pca_many
    from vip_amd.synth import synth_adi
    ctx = B.get_context()

    def lat(fn, reps, warm=1, settle=False):
        t_begin = time.perf_counter()
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        if not settle:
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3
        # Small calls straight after the chip-filling legs: for the first 0.1-0.25 s (one or two stretches of 40-160 calls, in
        # about every other process) a call takes 1.2-2.5 ms instead of 0.45, then 0.45 for good (NOTES round 6: not the reported
        # shader clock, not the way the host waits).  The warm-up therefore lasts at least 0.6 s AND until two consecutive batches of
        # 20 calls agree to 5 %; the figure is the median of the means of reps / 20 further batches.
        def batch():
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 20 * 1e3
        prev, t_end = None, t_begin + 4.0
        while time.perf_counter() < t_end:
            cur = batch()
            if prev is not None and abs(cur - prev) < 0.05 * prev and time.perf_counter() - t_begin > 0.6:
                break
            prev = cur
        return float(np.median([batch() for _ in range(max(3, reps // 20))]))

    def stages_of(fn):
        ctx.set_option("timing", 1)
        ctx.reset_timers()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        st = {s_: ctx.stage_ms(s_) / 3 for s_ in LEG_STAGES if ctx.stage_count(s_) > 0}
        ctx.set_option("timing", 0)
        return {k_: round(v_, 4) for k_, v_ in st.items()}

    out = {}
    # (a) the tutorial's shape
    n, N, k = 61, 101, 5
    cube, ang = synth_adi(n, N, seed=11)
    ct = torch.from_numpy(cube).cuda()
    pin_small = torch.empty((N, N), dtype=torch.float32).pin_memory()

    def resident_call():                 # (the frame into pinned memory, as the headline's steps: a pageable destination is bimodal
        fr = pca(ct, ang, ncomp=k, verbose=False, check_memory=False)      # here, 0.45 or 1.2 ms per call from run to run)
        pin_small.copy_(fr, non_blocking=False)      # (blocking copy: returns with the frame on the host; stream.synchronize() after an
                                                     #  asynchronous copy cost 0.7 ms more on some boxes, tools/sync_probe.py)
    ms_res = lat(resident_call, 100, warm=60, settle=True)
    ms_np = lat(lambda: pca(cube, ang, ncomp=k, verbose=False, check_memory=False), 100, warm=20)
    many = [ct] * 64
    pca_many(many[:4], [ang] * 4, ncomp=k, check_memory=False)
    t0 = time.perf_counter()
    outs = pca_many(many, [ang] * len(many), ncomp=k, check_memory=False)
    torch.cuda.synchronize()
    ms_many = (time.perf_counter() - t0) / len(many) * 1e3
    assert len(outs) == len(many)
    pub = 61 / 2.37
    out["tutorial_61x101x101_k5"] = {
        "metric": "ADI cube frames/sec at ncomp=5, 61x101x101 (one synchronous pca() call, cube resident, frame to the host)",
        "value": n / (ms_res * 1e-3), "unit": "frames/s", "latency_ms_per_call": ms_res,
        "value_numpy_in": n / (ms_np * 1e-3), "latency_ms_numpy_in": ms_np,
        "value_pipelined": n / (ms_many * 1e-3), "ms_per_cube_pipelined": ms_many,
        "stages_serial_ms": stages_of(resident_call),
        "vs_baseline": n / (ms_res * 1e-3) / pub, "vs_baseline_numpy_in": n / (ms_np * 1e-3) / pub,
        "baseline": {"value": pub, "unit": "frames/s", "source": "docs/source/tutorials/01A_quickstart.ipynb cell 53: "
                     "pca(cube 61x101x101, ncomp=5) 2.37 s (BASELINE.md section 1)",
                     "caveat": "an incidental notebook output on an unspecified laptop (17 GB RAM), not a benchmark: the same "
                               "shape and call, other hardware, other data (the tutorial's NACO cube; synthetic here)"}}
    del ct
    # (b) C2 with odd frames
    n, N, k = 400, 511, 20
    cube, ang = synth_adi(n, N, seed=12)
    ct = torch.from_numpy(cube).cuda()
    del cube
    pin_odd = torch.empty((N, N), dtype=torch.float32).pin_memory()

    def fn():
        fr = pca(ct, ang, ncomp=k, verbose=False, check_memory=False)
        pin_odd.copy_(fr, non_blocking=False)
    ms_odd = lat(fn, 8)
    out["odd_400x511x511_k20"] = {
        "metric": "ADI cube frames/sec at ncomp=20, 400x511x511 (one synchronous pca() call, cube resident)",
        "value": n / (ms_odd * 1e-3), "unit": "frames/s", "latency_ms_per_call": ms_odd, "stages_serial_ms": stages_of(fn),
        "vs_512_px": (ms_odd / c2_latency_ms if c2_latency_ms else None),
        "note": "Le = 2044 = 4 x 7 x 73: every real shift of the three shears as a circular convolution of 1024 points "
                "(derotate_conv.inc); vs_512_px = this call's latency over the 400x512x512 call's"}
    del ct
    torch.cuda.empty_cache()
    return out


def sharded_leg(mode, world, rank, backend, steps, warmup, frames=400, size=512, ncomp=20, spectrum=None):
    """One problem sharded over all ranks (strong scaling): every rank holds the same synthetic input in HBM; a step is
    one complete sharded call ending with the final frame on every rank.  Returns the record (every rank).
    (VIPMI_BENCH_BACKEND=gloo + VIPMI_BENCH_DEVICE=0 run the multi-rank code path on a single-GPU box.)"""
    import torch
    import torch.distributed as dist
    from vip_amd import dist as D
    from vip_amd.synth import synth_adi, synth_adi_device
    n, N, k = frames, size, ncomp
    if mode == "4d":
        nch, n, N = 39, 200, 256
        cube_t = torch.stack([synth_adi_device(n, N, seed=s)[0] for s in range(nch)])
        angles = np.linspace(0, 90, n)
        what = "configs[3]: %dx%dx%dx%d IFS cube, per-channel PCA ncomp=%d + spectral mean, channels sharded" % (nch, n, N, N, k)

        def step():
            return D.pca_4d(cube_t, angles, ncomp=k, verbose=False, check_memory=False)[0]
        units = nch * n
    elif mode == "annular":
        cube_t, angles = synth_adi_device(n, N, seed=0)
        what = "configs[2]: %dx%dx%d ADI cube, annular PCA (asize 32 -> %d annuli, ncomp=10), annuli sharded" % (n, N, N, N // 64)

        def step():
            return D.pca_annular(cube_t, angles, ncomp=10, asize=32, fwhm=4, delta_rot=(0.1, 1), n_segments=1)
        units = n
    elif mode == "single-cube":
        cube_t, angles = synth_adi_device(n, N, seed=0, **(dict(nmodes=spectrum[0], halving=spectrum[1]) if spectrum else {}))
        what = "%dx%dx%d ADI cube, full-frame PCA ncomp=%d, ONE cube sharded (Gram all-reduce + 2 all-to-all)" % (n, N, N, k)
        if spectrum:
            what += "; decaying speckle spectrum (%d modes, amplitude halves every %g modes) instead of the generator's 30 / 3" % spectrum

        def step():
            return D.pca_single_cube(cube_t, angles, k)
        units = n
    else:
        raise ValueError(mode)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, warmup)):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
        out = out if not hasattr(out, "cpu") else out.cpu()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # one more step with the phase clock on (every phase boundary synchronises: not part of the timed region): where the step's
    # time goes -- compute phases against the collectives -- MAX over the ranks per phase
    D.phase_timing(True)
    step()
    phases = D.phase_timing(False)
    if world > 1 and phases:
        names = sorted(phases)
        t = torch.tensor([phases[k_] for k_ in names], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        phases = dict(zip(names, [float(v) for v in t.tolist()]))
    # one more step with the library's stage timers on (hipEvent pairs around every stage and around the three shear kernels, on
    # the streams they run on; not part of the timed region): the leg's own roofline object
    leg_roof = None
    if world == 1:
        try:
            leg_roof = leg_roofline(mode, step, 1e3 * elapsed / steps, n, N, k, nch if mode == "4d" else 1)
        except Exception as e:                  # (never costs the leg)
            sys.stderr.write("leg roofline (%s) failed: %r\n" % (mode, e))
    # the frame's corners are NaN by design (mask_val of the vip-fft rotation): check the disk the rotation keeps
    out = np.asarray(out)
    c = out.shape[-1] // 2
    if not np.isfinite(out[c - 8:c + 8, c - 8:c + 8]).all():
        raise SystemExit("bench.py --mode %s: the final frame is not finite around its centre" % mode)
    del cube_t
    torch.cuda.empty_cache()
    from vip_amd import backend as _B
    ctx = _B.get_context()
    solver = {k_: int(ctx.get_option("eigh_fast_last_" + k_)) for k_ in ("reason", "products", "rounds", "locked")} if mode == "single-cube" else None
    return {"metric": "frames/sec, %s" % mode, "value": units * steps / elapsed, "unit": "frames/s",
            "eigh_fast_last": solver,          # the verified fast path's last run on this rank: reason 0 = converged
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "phases_ms": {k_: round(v_, 3) for k_, v_ in (phases or {}).items()},
            "roofline": leg_roof,
            "config": {"workload": what, "parallelism": "one problem over %d GPU(s), collectives of SURVEY 8(e)" % world}}


LEG_STAGES = ("scale", "gram", "eigh", "project", "derotate", "collapse", "k_rot_s1", "k_rot_s2", "k_rot_s3", "k_rot_aux")


def leg_pmc_traffic(tag, kernel="rs_shear2"):
    """HBM bytes per launch of a leg's dominant kernel from the round's committed PMC passes of THAT configuration
    (profiles/rNN_pmc_hbm_<tag>.json, tools/pmc_summary.py); None when no such pass is committed."""
    import glob
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_%s.json" % tag))
                   if re.fullmatch(r"r\d+_pmc_hbm_%s\.json" % tag, os.path.basename(f)))
    if not files:
        return None
    try:
        doc = json.load(open(files[-1]))
        for name, e in doc["kernels"].items():
            if name.startswith(kernel):
                return 1024.0 * (e["FETCH_SIZE_KB_per_launch"] + e["WRITE_SIZE_KB_per_launch"])
    except Exception:
        return None
    return None


def leg_roofline(mode, step, ms_per_step, n, N, k, nch):
    """The roofline object of one strong leg (one GPU): SURVEY 8(d)'s algorithmic bytes of the configuration over the measured
    step (`call_frac`), the per-stage times of one step (hipEvents of the library's stage timers, summed over the contexts of
    the streams the call uses), and the dominant kernel's own fraction: `rs_shear2*` -- the column shear of the FFT derotation, as
    at C2 -- with the derotation stage's algorithmic bytes (2 P 4 per frame) over its launch time; for the annular configuration
    the stage that takes most of the step is the batched library eigensolver, reported beside it with its nominal flops."""
    import torch
    from vip_amd import backend as B
    ctxs = B.all_contexts()
    for c in ctxs:
        c.set_option("timing", 1)
        c.reset_timers()
    step()
    torch.cuda.synchronize()
    st = {}
    for s_ in LEG_STAGES:
        ms = sum(max(c.stage_ms(s_), 0.0) for c in ctxs)
        cnt = sum(c.stage_count(s_) for c in ctxs)
        if cnt > 0:
            st[s_] = {"ms": ms, "launches": cnt}
    for c in ctxs:
        c.set_option("timing", 0)
    P = float(N * N)
    frames = n * nch
    if mode == "annular":
        # "bytes per annulus as above with P -> npx_ann" (Gram 1 + project 2 passes over the annuli's pixels)
        # + derotation 2 n P 4 + collapse n P 4
        from vip_amd.psfsub.pca_local import cached_annulus_plan
        plan, _ = cached_annulus_plan((N, N), np.linspace(0, 90, n), 0, 4, 32, 1, (0.1, 1), 10, 2, 200, 0)
        npx = float(sum(len(sg["pix"]) for sg in plan))               # 205,609 at C3
        call_bytes = 3.0 * n * npx * 4 + 3.0 * n * P * 4
        tag = "c3"
    else:
        call_bytes = 6.0 * frames * P * 4                     # 24 n P bytes per cube (per channel)
        tag = "c4" if mode == "4d" else "c5"
    roof = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "call_bytes": call_bytes,
            "call_frac": call_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "stages_ms": {k_: round(v_["ms"], 3) for k_, v_ in st.items()},
            "dominant_stage": (max((k_ for k_ in st if not k_.startswith("k_")), key=lambda k_: st[k_]["ms"]) if st else None)}
    if "k_rot_s2" in st:
        launches = st["k_rot_s2"]["launches"]
        dur_ms = st["k_rot_s2"]["ms"] / launches
        alg = 2.0 * P * 4 * frames / launches
        Le = 4 * N
        fft_flops = (frames / launches) * (Le / 2) * (2 * 5 * Le * np.log2(Le) + 16 * Le)
        roof.update({"kernel": "rs_shear2 (Le = %d)" % Le, "achieved": alg / (dur_ms * 1e-3) / 1e9,
                     "frac": alg / (dur_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": dur_ms, "launches_per_step": launches,
                     "flop_frac": fft_flops / (dur_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TF,
                     "traffic": leg_pmc_traffic(tag) if tag != "c3" else pmc_traffic(n / launches, N)})
        if "derotate" in st:
            roof["stage_frac"] = 2.0 * frames * P * 4 / (st["derotate"]["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    if mode == "annular" and "eigh" in st:
        # 8 annuli x n libraries of <= 200 frames: Householder tridiagonalisation 4/3 m^3 flops each (the k leading vectors add little)
        m = min(200, n)
        flops = 8.0 * n * (4.0 / 3.0) * m ** 3
        roof["eigh"] = {"kernel": "tri_eig_kernel (batched library eigensolver, float64 vector pipe)", "ms": st["eigh"]["ms"],
                        "flops": flops, "f64_valu_frac": flops / (st["eigh"]["ms"] * 1e-3) / 1e12 / 78.6}
    return roof


def sharded_mode(args, world, rank, backend):
    import torch.distributed as dist
    rec = sharded_leg(args.mode, world, rank, backend, args.steps, args.warmup, args.frames, args.size, args.ncomp)
    if rank == 0:
        print(json.dumps(rec))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def strong_scaling_legs(world, rank, backend):
    """The three shardings BASELINE.json's configs name, ONE problem over all ranks each (strong scaling): the whole C5
    cube (configs[4]), annular PCA of the C2 cube (configs[2]), the 4-D IFS cube (configs[3]).  A few steps each."""
    import torch
    from vip_amd import dist as D
    from vip_amd.psfsub import pca, pca_annular
    from vip_amd.synth import synth_adi
    out = {}
    # the sharded routines against the single-GPU calls on one small cube (the only place where the RCCL collectives of
    # the three partitions run on more than one device: the tests have gloo and one GPU)
    cube, ang = synth_adi(30, 128, seed=3)
    cube_t = torch.from_numpy(cube).cuda()
    chk = {}
    ref = pca(cube, ang, ncomp=4, verbose=False)
    chk["single_cube"] = float(np.nanmax(np.abs(D.pca_single_cube(cube_t, ang, 4).cpu().numpy() - ref)))
    ref = pca_annular(cube, ang, ncomp=3, asize=16, fwhm=4, verbose=False)
    chk["annular"] = float(np.nanmax(np.abs(D.pca_annular(cube_t, ang, ncomp=3, asize=16, fwhm=4).cpu().numpy() - ref)))
    c4 = np.stack([cube, cube[::-1] * 0.5, cube * 0.25])
    ref = pca(c4, ang, ncomp=4, verbose=False)
    got = D.pca_4d(torch.from_numpy(c4).cuda(), ang, ncomp=4, verbose=False, check_memory=False)[0]
    chk["ifs_4d"] = float(np.nanmax(np.abs(np.asarray(got.cpu() if hasattr(got, "cpu") else got) - ref)))
    out["selfcheck_max_abs_diff_vs_one_gpu"] = chk
    if max(chk.values()) > 1e-4:
        raise RuntimeError("sharded results deviate from the single-GPU path: %r" % chk)
    del cube_t
    for key, mode, kw, st in (("single_cube_c5", "single-cube", dict(frames=2000, size=1024, ncomp=50), 3),
                              ("single_cube_c5_decaying_spectrum", "single-cube",
                               dict(frames=2000, size=1024, ncomp=50, spectrum=(150, 10.0)), 3),
                              ("annular_c3", "annular", dict(frames=400, size=512, ncomp=10), 10),
                              ("ifs_4d_c4", "4d", dict(frames=200, size=256, ncomp=20), 10)):
        r = sharded_leg(mode, world, rank, backend, st, 1, **kw)
        out[key] = {"value": r["value"], "unit": "frames/s", "ms_per_step": r["ms_per_step"], "steps": st,
                    "workload": r["config"]["workload"], "eigh_fast_last": r["eigh_fast_last"], "roofline": r["roofline"],
                    "phases_ms": r["phases_ms"]}      # (one extra step, synchronised at every phase boundary; MAX over ranks)
    return out


def self_spawn(ngpus):
    """`python bench.py --gpus N` outside torchrun: re-run this command line under torch.distributed.run, one rank per
    GPU (RCCL).  Fails loudly when the box has fewer than N devices (VIPMI_BENCH_DEVICE, the single-GPU test rig, lifts
    that check: all ranks then share one device and VIPMI_BENCH_BACKEND must be gloo)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if "VIPMI_BENCH_DEVICE" not in os.environ and have < ngpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this box" % (ngpus, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--ncomp", type=int, default=20)
    ap.add_argument("--scaling", default=None,
                    help="matrix_scaling of the reference (None = its default; 'temp-mean' subtracts the per-pixel mean)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="(internal) time the CPU oracle in this GPU-free process and print its JSON object")
    ap.add_argument("--no-latency", action="store_true", help="skip the un-pipelined latency measurement")
    ap.add_argument("--no-numpy-in", action="store_true", help="skip the numpy-in (PCIe-inclusive) legs")
    ap.add_argument("--setup-burst", type=int, default=0,
                    help="untimed pipelined calls issued in one burst before the W warm-up steps (experiments; see the comment at its use)")
    ap.add_argument("--no-strong", action="store_true",
                    help="skip the strong-scaling legs (C5 single cube / C3 annular / C4 4-D sharded over the ranks)")
    ap.add_argument("--no-stage-timing", action="store_true",
                    help="do not record per-stage hipEvents inside the timed region (no roofline object)")
    ap.add_argument("--mode", default="survey", choices=["survey", "single-cube", "annular", "4d"],
                    help="survey (default, the BASELINE metric): one cube per GPU, no data-path collective, weak scaling; "
                         "single-cube / annular / 4d: ONE problem sharded over the GPUs with the collectives of SURVEY 8(e) "
                         "(vip_amd.dist.pca_single_cube / pca_annular / pca_4d), strong scaling")
    ap.add_argument("--sustained-seconds", type=float, default=10.0,
                    help="length of the sustained leg (pipelined calls back to back, power sampled): long enough for an outside "
                         "observer (rocm-smi, the driver's gpu_busy samples) to see the card busy")
    ap.add_argument("--no-shapes", action="store_true", help="skip the legs on the reference's own (odd-sized) shapes")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="independent pca() calls in flight (one torch stream each); 1 = strictly serial")
    args = ap.parse_args()

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.frames, args.size, args.ncomp)))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args.gpus)                   # does not return

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # (test hooks: VIPMI_BENCH_DEVICE pins every rank to one device and VIPMI_BENCH_BACKEND=gloo replaces RCCL, so that the
    # multi-rank code path can be exercised on a single-GPU box; the driver's runs use neither)
    torch.cuda.set_device(int(os.environ.get("VIPMI_BENCH_DEVICE", local_rank)))
    backend = os.environ.get("VIPMI_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))

    from vip_amd import backend as B
    from vip_amd.psfsub import pca
    from vip_amd.synth import synth_adi

    if args.mode != "survey":
        return sharded_mode(args, world, rank, backend)

    n, N, k = args.frames, args.size, args.ncomp
    depth = max(1, args.pipeline)
    # one DISTINCT cube per call in flight (survey mode: the calls are independent cubes, not one cube twice)
    cubes_t = []
    for d in range(depth):
        cube, angles = synth_adi(n, N, seed=rank * depth + d)
        cubes_t.append(torch.from_numpy(cube).cuda())
        del cube
    cube_t = cubes_t[0]
    ctx = B.get_context()

    frame_host = torch.empty((N, N), dtype=torch.float32).pin_memory()

    def step():
        # D2H of the final frame is part of the metric: into pinned memory, as the pipelined loop below does (a pageable
        # destination costs 0.14 ms more per 1 MB frame: tools/frame_d2h_probe.py); the blocking copy returns with the frame on the host
        frame = pca(cube_t, angles, ncomp=k, scaling=args.scaling, verbose=False, check_memory=False)
        frame_host.copy_(frame, non_blocking=False)
        return frame_host

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Pipelined issue: step i runs on stream i % depth.  Every step is still one complete pca() call ending
    # with the D2H of its frame (into pinned memory, non-blocking); nothing is skipped or cached -- the calls
    # are independent (one cube each, as in a survey / contrast-curve loop), so the latency-bound eigensolver
    # of one call overlaps the FFT derotation of the previous one.  All K frames are on the host when the
    # closing barrier returns.
    # (the library's own cached side streams -- the ones pca_many / the annular and 4-D fronts use: HIP maps every stream onto one of
    #  GPU_MAX_HW_QUEUES = 4 hardware queues, and with more streams than queues in a process two of them share one -- an upload on
    #  one stream then queues behind the kernels of the other: pca_many(numpy cubes) measured 15.2 instead of 8.4 ms per cube
    #  after this loop had run on two streams of its own, INTEGRATION.md "Threads, streams")
    streams = B.side_streams(depth)
    pinned = [torch.empty((N, N), dtype=torch.float32).pin_memory() for _ in range(max(args.steps, args.warmup, 1))]
    if depth > 1:
        B.set_async(True)       # (VIPMI_RESERVE_CUS=<n> keeps n CUs free of the shear kernels; measured best: 0)

    last = [None]
    step_events = None            # VIPMI_BENCH_TRACE=1: an event after every timed step (printed to stderr)

    def run(nsteps):
        if depth == 1:
            for i in range(nsteps):
                last[0] = step()
            return
        for i in range(nsteps):
            with torch.cuda.stream(streams[i % depth]):
                frame = pca(cubes_t[i % depth], angles, ncomp=k, scaling=args.scaling, verbose=False, check_memory=False)
                pinned[i % len(pinned)].copy_(frame, non_blocking=True)
                if step_events is not None:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    step_events.append(ev)

    STAGES = ("scale", "gram", "eigh", "project", "derotate", "collapse", "k_rot_s1", "k_rot_s2", "k_rot_s3",
              "k_rot_aux")
    timing = not args.no_stage_timing

    def set_timing(on):
        # inside the timed region only the roofline kernel (the column shear) carries a hipEvent pair (timing level 2);
        # timing every stage costs 2-3 % of the throughput and is done in the serial pass below instead
        for c in B.all_contexts():
            c.set_option("timing", 2 if on else 0)

    torch.cuda.synchronize()
    run(depth)                                  # creates the per-stream contexts
    torch.cuda.synchronize()
    # (experiments) one deep burst before the warm-up steps.  With FOUR staging slots per upload the host ran eight calls ahead
    # of the two streams, and the first time it did so in a process calls 2..10 took 6.2 ms instead of 4.9 (the HIP runtime
    # growing its per-queue pools: tools/pipe_history.py) -- 20 timed steps after 5 warm-up steps measured that transient
    # (108-111 ms) unless a burst of 20 had gone before (100 ms).  The library now keeps ONE slot (option upload_ring): the
    # host stays about one call ahead, the transient is gone (100.4 ms without any burst) and small cubes gained 30 %.
    if depth > 1 and args.setup_burst > 0:
        run(args.setup_burst)
        torch.cuda.synchronize()
    if timing:
        set_timing(True)                        # hipEvent pairs around every stage / shear kernel, on the
    run(args.warmup)                            # stream the kernels are launched on (events are created here)
    torch.cuda.synchronize()
    for c in B.all_contexts():
        c.reset_timers()
    # A full collection of the interpreter's cyclic GC walks the ~170 k objects that importing torch leaves tracked: 65 ms on
    # this host, once every few dozen calls -- 13 of the 5 ms steps timed here (tools/c3_small_kernels.py: one C3 call in fifteen
    # took 45-107 instead of 15 ms; none with the objects frozen).  As timeit does for its own loops, the collector is kept out
    # of the timed regions: everything alive now moves to the permanent generation, new garbage is still collected.
    gc.collect()
    gc.freeze()
    # timed region
    barrier()
    if os.environ.get("VIPMI_BENCH_TRACE") and depth > 1:
        step_events = []
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if step_events is not None:
        print("GPU end of each timed step (ms):", " ".join("%.1f" % ev0.elapsed_time(e) for e in step_events), file=sys.stderr)
        step_events = None
    if depth > 1:
        B.check_deferred()
    out = pinned[args.steps - 1] if depth > 1 else last[0]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(torch.isfinite(out).all())

    # per-stage / per-kernel durations of exactly the K timed steps, summed over the per-stream contexts
    roof = None
    stages = {}
    ms_svd = None
    if timing:
        for s_ in STAGES:
            ms = sum(max(c.stage_ms(s_), 0.0) for c in B.all_contexts())
            cnt = sum(c.stage_count(s_) for c in B.all_contexts())
            if cnt > 0:
                stages[s_] = {"ms_per_step": ms / args.steps, "launches_per_step": cnt / args.steps}
        set_timing(False)
        P = N * N
        if "k_rot_s2" in stages:
            launches = stages["k_rot_s2"]["launches_per_step"]
            dur_ms = stages["k_rot_s2"]["ms_per_step"] / launches
            frames_per_launch = n / launches
            alg_bytes = 2.0 * P * 4 * frames_per_launch          # derotate = 2*P*4 bytes per frame
            achieved = alg_bytes / (dur_ms * 1e-3) / 1e9
            roof_launches, roof_alg_bytes = launches, alg_bytes
            # FFT arithmetic of the kernel (informational: it is instruction-bound, not HBM-bound): per frame Le/2 pairs of
            # columns, each one forward + one inverse complex transform of Le points (5 Le log2 Le flop each) + the
            # two-for-one spectral mixing (2 complex multiply-adds per bin = 16 Le flop); Le = 4N
            Le = 4 * N
            fft_flops = frames_per_launch * (Le / 2) * (2 * 5 * Le * np.log2(Le) + 16 * Le)
            roof = {"bound": "hbm", "kernel": "rs_shear2", "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "flop_frac": fft_flops / (dur_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TF,
                    "flop_peak_tflops": FP32_VALU_PEAK_TF, "flops_per_launch": fft_flops,
                    "traffic": pmc_traffic(frames_per_launch, N),
                    "avg_launch_ms": dur_ms, "frames_per_launch": frames_per_launch,
                    "note": "duration = hipEvents around the kernel inside the timed region, where it shares the GPU "
                            "with the kernels of the other calls in flight (isolated_*: same kernel in a serial "
                            "call); the kernel is an FFT on the vector pipe that runs AT the board's power cap (`power`: "
                            "1380 of 1400 W under the shears, shader clock 1.8-1.96 of 2.4 GHz; DESIGN.md 4), so neither the "
                            "HBM nor the nominal vector peak is its ceiling"}

    if depth > 1:
        B.set_async(False)
    # un-pipelined latency of one call and the stage / kernel durations when a call has the GPU to itself
    latency_ms = None
    stages_serial = {}
    if not args.no_latency:
        torch.cuda.synchronize()
        step()
        # the latency itself WITHOUT the stage timers (timing = 1 brackets every stage and kernel with hipEvents: ~40 event
        # records per call, 0.3-0.6 ms on the box's host -- round 3 timed the call with them on); then three more calls with the
        # timers for the per-stage breakdown
        t1 = time.perf_counter()
        for _ in range(10):
            step()
        latency_ms = (time.perf_counter() - t1) / 10 * 1e3
        ctx.set_option("timing", 1)
        ctx.reset_timers()
        for _ in range(3):
            step()
        for s_ in STAGES:
            if ctx.stage_count(s_) > 0:
                stages_serial[s_] = ctx.stage_ms(s_) / 3
        ctx.set_option("timing", 0)
        # ms/SVD (BASELINE metric 2): scaling + Gram + eigensolver of one un-pipelined call
        ms_svd = sum(stages_serial[s_] for s_ in ("scale", "gram", "eigh") if s_ in stages_serial)
        if roof is not None and "k_rot_s2" in stages_serial:
            iso_ms = stages_serial["k_rot_s2"] / roof_launches
            roof["isolated_avg_launch_ms"] = iso_ms
            roof["isolated_frac"] = roof_alg_bytes / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if roof is not None and "derotate" in stages_serial:
            # the same bytes charged to the WHOLE derotation stage (its three shears + auxiliaries), and the call's
            # algorithmic bytes (Gram n P 4 + project 2 n P 4 + derotation 2 n P 4 + median n P 4) over a serial
            # call and over the pipelined step
            roof["stage_frac"] = 2.0 * n * P * 4 / (stages_serial["derotate"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            roof["call_bytes"] = 6.0 * n * P * 4
            roof["call_frac"] = roof["call_bytes"] / (latency_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            roof["call_frac_pipelined"] = roof["call_bytes"] / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS

    # steady state over >= 10 s of pipelined calls (the K timed steps above include the pipeline's fill and drain, and
    # K = 20 lasts 0.1 s; rounds 1-5 ran this leg for 1.5 s, too short for the driver's gpu_busy samples to see): reported beside
    # `value`, never instead of it; runs before the CPU baseline and the strong legs
    sustained = None
    power = None
    if depth > 1 and not args.no_latency:
        B.set_async(True)
        ns = min(len(pinned), max(args.steps, 1))
        reps = max(1, int(np.ceil(args.sustained_seconds / max(elapsed * ns / args.steps, 1e-3))))
        barrier()
        sampler = PowerSampler(torch.cuda.current_device()).start()
        t1 = time.perf_counter()
        for _ in range(reps):
            run(ns)
        barrier()
        ts = time.perf_counter() - t1
        power = sampler.stop("sustained leg (pipelined calls, %.1f s)" % ts)
        B.check_deferred()
        B.set_async(False)
        if world > 1:
            t = torch.tensor([ts], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ts = float(t.item())
        sustained = {"value": world * n * ns * reps / ts, "unit": "frames/s", "steps": ns * reps, "seconds": ts}

    shapes = None
    if rank == 0 and not args.no_latency and not args.no_shapes and (n, N, k) == (400, 512, 20):
        try:
            shapes = shape_legs(pca, B, torch, latency_ms)
        except Exception as e:                  # (never costs the headline line)
            sys.stderr.write("shape legs failed: %r\n" % (e,))
    numpy_in = None
    if rank == 0 and not args.no_latency and not args.no_numpy_in:
        try:
            numpy_in = numpy_in_legs(n, N, k, angles, rank * depth, pca, B, torch)
        except Exception as e:                  # (never costs the headline line)
            sys.stderr.write("numpy-in legs failed: %r\n" % (e,))
    del cubes_t, cube_t
    torch.cuda.empty_cache()
    rec = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * n * args.steps / elapsed
        rec = {
            "metric": "ADI cube frames/sec at ncomp=%d, %dx%dx%d" % (k, n, N, N) +
                      (" (%d independent pca() calls in flight, one distinct cube each; `value_serial` = n / wall of ONE pca() call, "
                       "the SURVEY 8(d) definition)" % depth if depth > 1 else ""),
            "value": value, "value_serial": (n / (latency_ms * 1e-3) if latency_ms else None), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "latency_ms_per_call": latency_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: %dx%dx%d ADI cube, full-frame PCA ncomp=%d, float32, "
                                   "vip-fft derotation, median collapse" % (n, N, N, k) +
                                   (", scaling=%s" % args.scaling if args.scaling else ""),
                       "cubes_per_step": world, "parallelism": "one cube per GPU (no data-path collective)",
                       "pipeline_depth": depth, "untimed_setup_burst": (args.setup_burst if depth > 1 else 0)},
            "ms_per_svd": ms_svd,
            "stages": stages,
            "stages_serial_ms": stages_serial,
            "roofline": roof,
            "sustained": sustained,
            "power": power,
            "h2d_ms": (numpy_in or {}).get("h2d_ms"),
            "value_numpy_in": (numpy_in or {}).get("value"),
            "numpy_in": numpy_in,
            "shapes": shapes,
            "strong": None,
        }
        if not args.no_cpu_baseline:
            # in a fresh process that never initialises the GPU runtime (the oracle forks a process pool)
            import subprocess
            env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--frames", str(n),
                                 "--size", str(N), "--ncomp", str(k)], capture_output=True, text=True, env=env)
            try:
                rec["cpu_baseline"] = json.loads(cp.stdout.strip().splitlines()[-1])
            except Exception:
                sys.stderr.write("cpu_baseline failed: %s\n" % cp.stderr[-2000:])
                rec["cpu_baseline"] = None

    # Strong-scaling legs LAST, under a watchdog: the record above is complete, so a collective that fails or hangs on
    # some rank costs the `strong` object, never the headline line (still exactly one JSON line on rank 0).
    def emit_and_exit(err):
        if rank == 0:
            rec["strong"] = {"error": err}
            print(json.dumps(rec))
            sys.stdout.flush()
        os._exit(0)

    if not args.no_strong and (n, N, k) == (400, 512, 20):
        import threading
        if world > 1:
            dist.barrier()                      # (rank 0 has just spent ~30 s in the CPU baseline)
        done = threading.Event()
        limit = float(os.environ.get("VIPMI_BENCH_STRONG_LIMIT_S", "420"))

        def dog():
            if not done.wait(limit):
                emit_and_exit("strong-scaling legs did not finish within %.0f s" % limit)
        threading.Thread(target=dog, daemon=True).start()
        try:
            strong = strong_scaling_legs(world, rank, backend)
        except BaseException as e:              # the other ranks may be stuck in a collective: no barrier, just leave
            sys.stderr.write("strong-scaling legs failed on rank %d: %r\n" % (rank, e))
            emit_and_exit(repr(e)[:300])
        done.set()
        if rank == 0:
            rec["strong"] = strong
    if rank == 0:
        print(json.dumps(rec))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
