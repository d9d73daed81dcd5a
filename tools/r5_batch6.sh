#!/bin/bash
O=gpurun_out/r5f; mkdir -p $O
timeout 900 python -X faulthandler -m pytest tests/test_gpu_pca.py tests/test_gpu_kernels.py -q -x -s -k "float64 or 6144 or few_vectors" 2>&1 | grep -v amdgpu.ids | tail -30 > $O/pytest.txt
timeout 600 python bench.py --no-cpu-baseline --no-strong > $O/bench.json 2> $O/bench.err
cat $O/pytest.txt; python - <<'P'
import json
r = json.load(open("gpurun_out/r5f/bench.json"))
for k in ("value", "value_serial", "ms_per_step", "latency_ms_per_call", "ms_per_svd", "h2d_ms", "value_numpy_in", "power", "numpy_in", "sustained"):
    print(k, r.get(k))
P
nproc; uptime
