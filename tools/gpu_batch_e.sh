#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3 -o c3 -- python tools/time_c3.py > gpurun_out/prof_c3.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_c3/**/*kernel_stats.csv', recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:16]:
    print("%-90s calls %6s total_ms %9.3f avg_us %9.1f pct %s" % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
