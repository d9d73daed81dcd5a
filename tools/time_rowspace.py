"""(I - C) A of annular PCA: the (n x n) x (n x npx) product through the row-space kernel, per annulus size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vip_amd import backend as B
ctx = B.get_context()
n = 400
for npx in (3205, 9644, 16064, 28940, 48028):
    C = torch.randn(n, n, device="cuda") * 0.05
    A = torch.randn(n, npx, device="cuda")
    R = torch.empty(n, npx, device="cuda")
    for _ in range(2): ctx.call("vipmi_rowspace_gemm_f32", B.ptr(C), B.ptr(A), n, n, npx, None, B.ptr(R))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ctx.call("vipmi_rowspace_gemm_f32", B.ptr(C), B.ptr(A), n, n, npx, None, B.ptr(R))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    ref = C.double() @ A.double()
    err = float((R.double() - ref).abs().max() / ref.abs().max())
    print("npx=%d: %.3f ms  %.1f TF/s  (rel err %.1e)" % (npx, ms, 2.0 * n * n * npx / ms / 1e9, err))
