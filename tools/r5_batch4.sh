#!/bin/bash
O=gpurun_out/r5d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pca.py -q -x -s -k "float64 or 6144" 2>&1 | grep -v amdgpu.ids | tail -15 > $O/pytest_f64.txt
timeout 600 python bench.py --no-cpu-baseline --no-strong > $O/bench.json 2> $O/bench.err
cat $O/pytest_f64.txt; python - <<'P'
import json
r = json.load(open("gpurun_out/r5d/bench.json"))
for k in ("value", "value_serial", "ms_per_step", "latency_ms_per_call", "ms_per_svd", "h2d_ms", "value_numpy_in", "power", "numpy_in", "sustained", "stages_serial_ms"):
    print(k, r.get(k))
print({k: v for k, v in r["roofline"].items() if k != "note"})
P
tail -5 $O/bench.err
