"""The pipelined loop of bench.py with an event after every call: when each call ends on the GPU and when the host returned from
issuing it (un-profiled: looks for the stalls that only show without rocprofv3).   python tools/pipe_steps.py [steps] [gc=0|1]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi_device
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
use_gc = not (len(sys.argv) > 2 and sys.argv[2] == "gc=0")
n, N, k, depth = 400, 512, 20, 2
cubes = [synth_adi_device(n, N, seed=s)[0] for s in range(depth)]
ang = np.linspace(0, 90, n)
streams = [torch.cuda.Stream() for _ in range(depth)]
pinned = [torch.empty((N, N), dtype=torch.float32).pin_memory() for _ in range(K)]
B.set_async(True)
def run(m, rec=None):
    for i in range(m):
        with torch.cuda.stream(streams[i % depth]):
            fr = pca(cubes[i % depth], ang, ncomp=k, verbose=False, check_memory=False)
            pinned[i % K].copy_(fr, non_blocking=True)
            if rec is not None:
                e = torch.cuda.Event(enable_timing=True); e.record(); rec.append((e, time.perf_counter()))
run(depth); torch.cuda.synchronize(); run(5); torch.cuda.synchronize()
TIM = len(sys.argv) > 3 and sys.argv[3] == 'timing'
for rep in range(4):
    if TIM:
        for c in B.all_contexts(): c.set_option('timing', 1); c.reset_timers()
    if not use_gc: gc.collect(); gc.disable()
    rec = []
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e0.record(); t0 = time.perf_counter()
    run(K, rec)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) * 1e3
    if not use_gc: gc.enable()
    ends = [e0.elapsed_time(e) for e, _ in rec]; host = [(t - t0) * 1e3 for _, t in rec]
    print("rep %d: elapsed %.2f ms; GPU end of each call: %s" % (rep, el, " ".join("%.1f" % x for x in ends)))
    print("        host returned from issuing:      %s" % " ".join("%.1f" % x for x in host))
    if TIM:
        print('        stage ms per call:', {st: round(sum(max(c.stage_ms(st), 0) for c in B.all_contexts()) / K, 3) for st in ('gram', 'eigh', 'project', 'derotate', 'collapse', 'k_rot_s1', 'k_rot_s2', 'k_rot_s3')})
B.check_deferred(); B.set_async(False)
