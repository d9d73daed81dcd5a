"""Pipelined (asynchronous, two streams) throughput on small cubes for staging-ring depths (option upload_ring)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.psfsub import pca
from vip_amd.synth import synth_adi_device
depth = 2
streams = [torch.cuda.Stream() for _ in range(depth)]
B.set_async(True)
for (n, N, k) in ((50, 128, 5), (30, 64, 3), (100, 256, 10), (400, 512, 20)):
    cubes = [synth_adi_device(n, N, seed=s)[0] for s in range(depth)]
    ang = np.linspace(0, 90, n)
    pinned = torch.empty((N, N), dtype=torch.float32).pin_memory()
    for ring in (4, 2, 1):
        def run(m):
            for i in range(m):
                with torch.cuda.stream(streams[i % depth]):
                    B.get_context().set_option("upload_ring", ring)
                    fr = pca(cubes[i % depth], ang, ncomp=k, verbose=False, check_memory=False)
                    pinned.copy_(fr, non_blocking=True)
        # (the ring is sized at first use: a fresh name per setting would be needed to resize it -- so each setting runs in
        # its own process; see the shell loop)
        ring = int(os.environ.get("RING", "4"))
        run(40); torch.cuda.synchronize()
        M = 400 if n < 200 else 60
        t0 = time.perf_counter(); run(M); torch.cuda.synchronize(); el = time.perf_counter() - t0
        print("ring %d  %dx%dx%d k=%d: %.1f us per call" % (ring, n, N, N, k, el / M * 1e6))
        break
B.check_deferred(); B.set_async(False)
