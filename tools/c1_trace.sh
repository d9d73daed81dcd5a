#!/bin/bash
# kernel trace of back-to-back C1 calls (50 x 128 x 128, k = 5): per-kernel durations and the gaps between them
cd /tmp && export TMPDIR=/tmp
cat > /tmp/c1.py <<'PY'
import sys, os; sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, numpy as np
from vip_amd.synth import synth_adi
from vip_amd.psfsub import pca
cube, ang = synth_adi(50, 128, 0); ct = torch.from_numpy(cube).cuda()
for _ in range(30): pca(ct, ang, ncomp=5, verbose=False, check_memory=False)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d /tmp/c1t -o k -- python /tmp/c1.py > /tmp/c1.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/c1t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last call: find the last occurrence of the collapse kernel and walk back to the previous one
idx = [i for i, r in enumerate(rows) if "median_kernel" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]["Start_Timestamp"]); prev_end = None
tot_k = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    nm = r["Kernel_Name"].replace("vipmi::", "").replace("(anonymous namespace)::", "")[:60]
    print("%7.1f us  +%5.1f gap  %6.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, nm))
    prev_end = e; tot_k += e - s
print("call span %.1f us, kernel time %.1f us, %d launches" % ((prev_end - t0) / 1e3, tot_k / 1e3, b - a))
PY
