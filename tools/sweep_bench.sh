#!/bin/bash
# Run ON THE GPU BOX: sweep of the pipelining knobs of bench.py (frames/s).
for depth in 2 3; do for rc in 0 8 16 24 32; do
  v=$(VIPMI_RESERVE_CUS=$rc timeout 200 python bench.py --pipeline $depth --no-cpu-baseline --no-latency --no-stage-timing --steps 30 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']))")
  echo "depth=$depth reserve_cus=$rc  $v frames/s"
done; done
