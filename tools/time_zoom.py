"""Time the mSDI zoom (batched A.B^T MFMA kernel) alone: 39 channels x 200 frames, 256 -> padded size and back."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd.preproc.rescaling import channel_operators, zoom_frames
z, n, N = 39, 200, 256
sc = np.linspace(1.0, 1.3, z)[::-1].copy()
x = torch.randn(n * z, N, N, device="cuda")
E = channel_operators(N, sc)
big = E.shape[1]
idx = np.tile(np.arange(z), n)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
ms = t(lambda: zoom_frames(x, E, idx))
fl = n * z * (2 * 2 * big * N * N + 2 * 2 * big * N * big)
print("forward zoom %d -> %d: %.1f ms  (%.1f TF/s)" % (N, big, ms, fl / ms / 1e9))
y = zoom_frames(x, E, idx)
Einv = channel_operators(big, sc, inverse=True, out_size=N)
ms = t(lambda: zoom_frames(y, Einv, idx))
ys = zoom_frames(y, Einv, idx).shape[1]
fl = n * z * (2 * 2 * ys * big * big + 2 * 2 * ys * big * ys)
print("inverse zoom %d -> %d: %.1f ms  (%.1f TF/s)" % (big, ys, ms, fl / ms / 1e9))
