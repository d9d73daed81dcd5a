"""Does a pageable host->device copy on one stream run beside kernels of another stream?  (pca_many with numpy cubes)"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
from vip_amd.synth import synth_adi
cube, ang = synth_adi(400, 512, 0)
ct = torch.from_numpy(cube).cuda()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
pin = torch.from_numpy(cube).pin_memory()
def t(fn, reps=4):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def up_page():
    with torch.cuda.stream(sA): return torch.from_numpy(cube).to("cuda")
def up_pin():
    with torch.cuda.stream(sA): return pin.to("cuda", non_blocking=True)
def rot(k=2):
    with torch.cuda.stream(sB):
        for _ in range(k): B.derotate(ct, ang)
print("upload pageable alone %.2f ms; pinned alone %.2f ms; 2 derotations alone %.2f ms" % (t(up_page), t(up_pin), t(rot)))
def both_page():
    rot(); up_page()
def both_pin():
    rot(); up_pin()
print("2 derotations (stream B) then pageable upload (stream A): %.2f ms; with pinned upload: %.2f ms" % (t(both_page), t(both_pin)))
# staging through a pinned buffer with a thread pool (numpy copies release the GIL)
from concurrent.futures import ThreadPoolExecutor
stage = torch.empty(cube.shape, dtype=torch.float32).pin_memory(); sv = stage.numpy()
for nthr in (4, 8, 16):
    pool = ThreadPoolExecutor(nthr)
    def stage_copy():
        chunks = np.array_split(np.arange(cube.shape[0]), nthr)
        list(pool.map(lambda idx: np.copyto(sv[idx[0]:idx[-1] + 1], cube[idx[0]:idx[-1] + 1]), chunks))
    t0 = time.perf_counter(); [stage_copy() for _ in range(3)]; ms = (time.perf_counter() - t0) / 3 * 1e3
    print("host staging copy into pinned memory, %d threads: %.2f ms (%.1f GB/s)" % (nthr, ms, cube.nbytes / ms / 1e6))
