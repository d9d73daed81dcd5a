"""Where does pca_many(numpy cubes) spend its wall time?  host timestamps around every upload and enqueue"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi
hosts = [synth_adi(400, 512, seed=s)[0] for s in range(3)]; ang = synth_adi(8, 16, 0)[1]; ang = np.linspace(0, 90, 400)
streams = B.side_streams(2)
def run(n_items, tag, sync_each=False):
    B.set_async(True)
    cur = torch.cuda.current_stream(); outs = []; log = []
    t00 = time.perf_counter()
    for i in range(n_items):
        st = streams[i % 2]; st.wait_stream(cur)
        with torch.cuda.stream(st):
            t0 = time.perf_counter(); t = B.to_device_f32(hosts[i % 3]); t1 = time.perf_counter()
            outs.append(pca(t, ang, ncomp=20, verbose=False, check_memory=False)); t2 = time.perf_counter()
            log.append((t1 - t0, t2 - t1))
    for st in streams: st.synchronize()
    B.check_deferred(); B.set_async(False)
    tot = time.perf_counter() - t00
    print(tag, "total %.1f ms = %.2f ms per cube; upload ms:" % (tot * 1e3, tot * 1e3 / n_items), " ".join("%.1f" % (a * 1e3) for a, b in log),
          "| enqueue ms:", " ".join("%.1f" % (b * 1e3) for a, b in log), flush=True)
if "after-pipe" in sys.argv:          # what bench.py has done before its numpy-in legs: resident cubes pipelined on two OTHER streams
    from vip_amd.synth import synth_adi_device
    cts = [synth_adi_device(400, 512, seed=s)[0] for s in range(2)]
    own = [torch.cuda.Stream() for _ in range(2)]
    pinned = [torch.empty((512, 512)).pin_memory() for _ in range(50)]
    B.set_async(True)
    for rep in range(3):
        for i in range(50):
            with torch.cuda.stream(own[i % 2]):
                fr = pca(cts[i % 2], ang, ncomp=20, verbose=False, check_memory=False)
                pinned[i].copy_(fr, non_blocking=True)
        torch.cuda.synchronize()
    B.check_deferred(); B.set_async(False)
    for _ in range(5): pca(cts[0], ang, ncomp=20, verbose=False, check_memory=False).cpu()
    if "drop" in sys.argv: del cts; torch.cuda.empty_cache()
    print("contexts alive:", len(B.all_contexts()))
run(3, "warm")
run(10, "pca_many-like")
# the same with the upload issued from pinned memory (truly asynchronous copy)
pins = [torch.from_numpy(h).pin_memory() for h in hosts]
orig = B.to_device_f32
B.to_device_f32 = lambda x, device=None: (x.to("cuda", non_blocking=True) if isinstance(x, torch.Tensor) and x.is_pinned() else orig(x, device))
hosts = pins
run(10, "pinned sources  ")
