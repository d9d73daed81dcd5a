#!/bin/bash
O=gpurun_out/r5m; mkdir -p $O
for opt in "" "eigh_wave_async=1" "eigh_wave_async=2" "" "eigh_wave_async=1"; do
  VIPMI_OPTS=$opt timeout 300 python bench.py --no-cpu-baseline --no-strong --no-latency --steps 100 2> /dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$opt', 'value %.0f  ms_per_step %.3f' % (r['value'], r['ms_per_step']), {k: round(v['ms_per_step'], 3) for k, v in r['stages'].items()})
" >> $O/wave_async.txt
done
cat $O/wave_async.txt
