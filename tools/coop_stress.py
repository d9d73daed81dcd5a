"""Several host threads, each on its own stream, issuing the lone n = 400 eigenproblem at the same time: the one-XCD layout of
tri_multi_kernel (32 whole-CU workgroups = a whole XCD) and the wave kernel.  Before CoopOrder (common.h) two launches that landed
on the same XCD, each resident in part, waited for each other until the barrier time-out.  python tools/coop_stress.py [threads reps]"""
import sys, os, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
nthr, reps = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 60)
rng = np.random.default_rng(1)
M = rng.standard_normal((400, 900)); G = M @ M.T
for wave in (0, 1):
    c0 = B.get_context(); c0.set_option("eigh_wave", wave); c0.set_option("eigh_fast", 0)
    ev0, ec0 = B.eigh_topk(torch.from_numpy(G.copy()).cuda(), 15); ev0, ec0 = ev0.cpu().numpy(), ec0.cpu().numpy()
    bad, errs = [0], []
    def work(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                c = B.get_context(); c.set_option("eigh_wave", wave); c.set_option("eigh_fast", 0); c.set_option("eigh_one_xcd", 1)
                for r in range(reps):
                    ev, ec = B.eigh_topk(torch.from_numpy(G.copy()).cuda(), 15)
                    if not (np.array_equal(ev.cpu().numpy(), ev0) and np.array_equal(ec.cpu().numpy(), ec0)): bad[0] += 1
                st = c.lib.vipmi_check_deferred(c.handle)
                if st: errs.append("deferred %d" % st)
                c.set_option("eigh_one_xcd", -1); c.set_option("eigh_wave", 1); c.set_option("eigh_fast", 1)
        except Exception as e:
            errs.append(repr(e))
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nthr)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print("eigh_wave=%d: %d threads x %d calls in %.2f s (%.2f ms per call overall), mismatches %d, errors %s" % (
        wave, nthr, reps, dt, 1e3 * dt / (nthr * reps), bad[0], errs), flush=True)
    c0.set_option("eigh_wave", 1); c0.set_option("eigh_fast", 1)
